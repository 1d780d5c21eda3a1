"""Timing of the split-bf16 GEMM at the 1x1-convolution and attention shapes of the model (B=16)."""
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
for cin, cout, hw in [(128, 256, 128), (256, 512, 64), (512, 256, 128), (256, 128, 256), (512, 512, 32)]:
    x = torch.randn(B, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    dy = torch.randn(B, cout, hw, hw, device='cuda')
    fl = 2.0 * B * hw * hw * cin * cout
    byt = 4.0 * B * hw * hw * (cin + cout)
    tf = timeit(lambda: ops.conv1x1_lowp(x, w, None, None))
    td = timeit(lambda: ops.conv1x1_lowp_dgrad(dy, w))
    tw = timeit(lambda: ops.conv1x1_lowp_wgrad(x, dy))
    print(f'1x1 {cin:4d}->{cout:4d} @{hw:3d}^2  fwd {tf:6.3f} ms {fl/tf/1e9:6.1f} TF {byt/tf/1e6:7.0f} GB/s | dgrad {td:6.3f} ms {byt/td/1e6:7.0f} GB/s | wgrad {tw:6.3f} ms {byt/tw/1e6:7.0f} GB/s', flush=True)
q = torch.randn(B, 512, 1024, device='cuda'); k = torch.randn_like(q); v = torch.randn_like(q)
t = timeit(lambda: ops.attention(q, k, v, 512 ** -0.5))
print(f'attention fwd B=16 C=512 T=1024: {t:6.3f} ms {2*2.0*B*1024*1024*512/t/1e9:6.1f} TF')
