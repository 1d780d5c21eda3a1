"""GroupNorm(32)+SiLU forward / backward at the model's activation shapes (B=16): achieved HBM GB/s
against the algorithmic traffic (fwd: read x twice + write y; bwd: read x, dy twice + write dx)."""
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
for c, hw in [(128, 256), (256, 128), (512, 64), (512, 32), (256, 256), (512, 128)]:
    x = torch.randn(B, c, hw, hw, device='cuda', requires_grad=True)
    g = torch.randn(c, device='cuda', requires_grad=True); b = torch.randn(c, device='cuda', requires_grad=True)
    dy = torch.randn(B, c, hw, hw, device='cuda')
    nbytes = 4.0 * B * c * hw * hw
    tf = timeit(lambda: ops.group_norm_silu(x.detach(), g.detach(), b.detach()))
    y = ops.group_norm_silu(x, g, b)
    tb = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True))
    print(f'GN {c:4d} @{hw:3d}^2  fwd {tf:6.3f} ms {3*nbytes/tf/1e6:6.0f} GB/s (3 passes) | bwd {tb:6.3f} ms {5*nbytes/tb/1e6:6.0f} GB/s (5 passes)', flush=True)
