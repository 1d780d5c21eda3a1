import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
x = torch.randn(16, 128, 256, 256, device='cuda'); y = torch.empty_like(x); z = torch.empty_like(x)
nb = x.numel() * 4
t = timeit(lambda: y.copy_(x)); print(f'copy        {2*nb/t/1e6:7.0f} GB/s')
t = timeit(lambda: torch.add(x, y, out=z)); print(f'add (2r+1w) {3*nb/t/1e6:7.0f} GB/s')
t = timeit(lambda: x.sum()); print(f'sum (1r)    {nb/t/1e6:7.0f} GB/s')
t = timeit(lambda: torch.mul(x, 2.0, out=y)); print(f'scale       {2*nb/t/1e6:7.0f} GB/s')
