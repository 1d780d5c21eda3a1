"""GroupNorm + SiLU -> MX operand records with the statistics supplied by the producing convolution (the step's common case:
one pass, 4 B read + 4 B written per element): achieved GB/s at the model's shapes (B = 16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops, lib as L

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

lib = L.load()
B = 16
for c, hw in [(128, 256), (256, 128), (512, 64), (256, 256), (512, 128)]:
    x = torch.randn(B, c, hw, hw, device='cuda')
    g = torch.randn(c, device='cuda'); b = torch.randn(c, device='cuda')
    xd = x.double().view(B, 32, -1)
    sums = torch.stack([xd.sum(-1), (xd * xd).sum(-1)], -1).contiguous()
    pa = torch.empty_like(x)
    stats = torch.empty(B * 32 * 2, device='cuda')
    ws = ops.workspace(lib.eovae_gn_workspace_bytes(B, c, hw * hw, 32), x.device)
    run = lambda: L.check(lib.eovae_gn_silu_fwd_packed_mx(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(pa), L.ptr(stats), L.ptr(sums), B, c,
                                                            hw * hw, 32, 1e-6, 1, L.ptr(ws), L.stream()), 'gn')
    t = timeit(run)
    nbytes = 8.0 * B * c * hw * hw
    print(f'GN apply -> MX records {c:4d} @{hw:3d}^2: {t:6.3f} ms {nbytes / t / 1e6:6.0f} GB/s', flush=True)
