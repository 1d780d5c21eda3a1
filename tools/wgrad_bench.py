"""A/B of the two f16c8 weight-gradient kernels at the real layer shapes (B=16), interleaved rounds in one process:
EOVAE_C8_WGRAD_W4=0 -> wgrad3x3_f16c8_kernel (8 waves), =1 -> conv_f16c8_wgrad_w4.hip (4 waves, all 9 taps per wave)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops, lib as L

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

shapes = [(128, 128, 256), (256, 256, 128), (512, 512, 64), (256, 128, 256), (512, 256, 128), (512, 512, 32)]
B = 16
lib = L.load()
ops.CONV_PRECISION = 'f16c8'
for cin, cout, hw in shapes:
    x = torch.randn(B, cin, hw, hw, device='cuda')
    dy = torch.randn(B, cout, hw, hw, device='cuda') * 1e-4
    dwp = torch.zeros(9 * cout * cin, device='cuda')
    db = torch.zeros(cout, device='cuda')
    amax = ops.absmax_bits(dy)
    flops = 2.0 * B * hw * hw * cin * cout * 9
    xrec = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True).detach()
    cases = {
        'x records': lambda: lib.eovae_conv3x3_f16c8_wgrad(L.ptr(xrec), L.ptr(dy), L.ptr(dwp), L.ptr(db), L.ptr(amax), None, B, cin, hw, hw, cout, hw, hw, 0, 0, 1, L.stream()),
        'x fp32   ': lambda: lib.eovae_conv3x3_f16c8_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), L.ptr(db), L.ptr(amax), None, B, cin, hw, hw, cout, hw, hw, 0, 0, 0, L.stream()),
    }
    best = {(k, v): 1e9 for k in cases for v in '01'}
    for rnd in range(3):
        for k, f in cases.items():
            for v in '01':
                os.environ['EOVAE_C8_WGRAD_W4'] = v
                best[(k, v)] = min(best[(k, v)], timeit(f))
    for k in cases:
        t0, t1 = best[(k, '0')], best[(k, '1')]
        print(f'{cin:4d}->{cout:4d} @{hw:3d}^2 wgrad {k}: 8-wave {t0:7.3f} ms {flops/t0/1e9:6.1f} TF | w4 {t1:7.3f} ms {flops/t1/1e9:6.1f} TF | x{t0/t1:.3f}', flush=True)
