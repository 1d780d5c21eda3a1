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
B=16
for cin, cout in [(3,128),(12,128),(128,3),(128,12)]:
    x = torch.randn(B, cin, 256, 256, device='cuda', requires_grad=(cin>12))
    w = (torch.randn(cout, cin, 3, 3, device='cuda')/ (cin*9)**0.5).requires_grad_(True)
    b = torch.zeros(cout, device='cuda', requires_grad=True)
    y = ops.conv2d(x, w, b, None, 'normal')
    gy = torch.randn_like(y)
    tf = timeit(lambda: ops.conv2d(x.detach(), w.detach(), b.detach(), None, 'normal'))
    inputs = (x, w, b) if x.requires_grad else (w, b)
    tb = timeit(lambda: torch.autograd.grad(y, inputs, gy, retain_graph=True))
    fl = 2.0*B*65536*cin*cout*9
    print(f'{cin:4d}->{cout:4d} @256^2 fwd {tf:6.3f} ms ({fl/tf/1e9:6.1f} TF alg) | bwd {tb:6.3f} ms')
