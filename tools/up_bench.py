"""A/B the upsample-conv layers: sub-pixel form vs plain (virtual-upsample) form, same device."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

B = 16
for c, hw in [(512, 32), (512, 64), (256, 128)]:
    x = torch.randn(B, c, hw, hw, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    dy = torch.randn(B, c, 2 * hw, 2 * hw, device='cuda')
    flops = 2.0 * B * 4 * hw * hw * c * c * 9
    for sub in (True, False):
        ops.SUBPIX_UPSAMPLE = sub
        if sub:
            tf = timeit(lambda: ops.up_conv_subpix(x, w, None))
            td = timeit(lambda: ops.up_conv_subpix_dgrad(dy, w, hw, hw))
            tw = timeit(lambda: ops.up_conv_subpix_wgrad(x, dy, w))
        else:
            tf = timeit(lambda: ops.conv3x3_lowp(x, w, None, None, c, ops.MODE_UP, False))
            td = timeit(lambda: ops.conv3x3_lowp(dy, w, None, None, c, ops.MODE_NORMAL, True))
            def wg():
                dwp = torch.zeros(9 * c * c, device='cuda')
                ops.L.load().eovae_conv3x3_bf16_wgrad(ops.L.ptr(x), ops.L.ptr(dy), ops.L.ptr(dwp), None, B, c, hw, hw, c, 2*hw, 2*hw, ops.MODE_UP, 3, 0, ops.L.stream())
            tw = timeit(wg)
        print(f'{c}ch {hw}->{2*hw} subpix={sub}: fwd {tf:.3f} ms ({flops/tf/1e9:.0f} TF alg) dgrad {td:.3f} ms ({flops/td/1e9:.0f}) wgrad {tw:.3f} ms ({flops/tw/1e9:.0f})', flush=True)
