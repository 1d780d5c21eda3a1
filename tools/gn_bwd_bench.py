"""GroupNorm backward at the model's layer shapes (B=16): one-pass cooperative kernel vs the two-pass kernels, same box.
    python tools/gn_bwd_bench.py            ms per call (all launches of the call), GB/s of the 12 (16 with the skip
                                            gradient) algorithmic bytes per element"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import lib as L

lib = L.load()
B = 16
shapes = [(128, 256), (256, 256), (256, 128), (512, 128), (512, 64), (128, 128)]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for c, hw in shapes:
    x = torch.randn(B, c, hw, hw, device='cuda')
    dy = torch.randn_like(x)
    add = torch.randn_like(x)
    g = torch.ones(c, device='cuda'); b = torch.zeros(c, device='cuda')
    y = torch.empty_like(x); dx = torch.empty_like(x)
    dg, db = torch.empty_like(g), torch.empty_like(b)
    stats = torch.empty(B * 32 * 2, device='cuda')
    bits = torch.zeros(1, dtype=torch.int32, device='cuda')
    ws = torch.empty(lib.eovae_gn_workspace_bytes(B, c, hw * hw, 32), dtype=torch.uint8, device='cuda')
    L.check(lib.eovae_gn_silu_fwd(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(stats), None, B, c, hw * hw, 32, 1e-6, 1,
                                  L.ptr(ws), L.stream()), 'fwd')
    for skip in (False, True):
        def call():
            L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(stats), L.ptr(add) if skip else None,
                                          L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(bits), B, c, hw * hw, 32, 1, L.ptr(ws),
                                          L.stream()), 'bwd')
        t = {}
        for rnd in range(3):
            for on in (0, 2):
                lib.eovae_gn_bwd_fused(on)
                t[min(on, 1)] = min(t.get(min(on, 1), 1e9), timeit(call))
        nb = x.numel() * (16 if skip else 12)
        print(f'{c:4d} ch @{hw:3d}^2 skip={int(skip)}: two-pass {t[0]:7.3f} ms | one-pass {t[1]:7.3f} ms '
              f'{nb / t[1] / 1e6:7.0f} GB/s | x{t[0] / t[1]:.3f}', flush=True)
    del x, dy, add, y, dx
