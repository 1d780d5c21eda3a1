"""Same-box A/B of the x2-upsample forward: conv_f16c8_q16up.hip on MX records of x (incl. its records pass) against the 8-wave
kernel's sub-pixel mode on the fp32 x, at the decoder's three layers (B = 16), alternating, HIP events."""
import sys
import torch
sys.path.insert(0, '.')
from eo_vae_amd import lib as L, ops
ops.CONV_PRECISION = 'f16c8'
lib = L.load()
for n, c, hw in [(16, 512, 32), (16, 512, 64), (16, 256, 128)]:
    torch.manual_seed(hw)
    x = torch.randn(n, c, hw, hw, device='cuda') * 2
    wt = torch.randn(c, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    b = torch.randn(c, device='cuda')
    res = {0: [], 1: []}
    resg = {0: [], 1: []}
    dy = torch.randn(n, c, 2 * hw, 2 * hw, device='cuda') * 1e-4
    for rep in range(6):
        for q in (0, 1):
            lib.eovae_set_option(b'c8_up_q16', q)
            with torch.no_grad():
                for _ in range(2):
                    ops.conv2d(x, wt, b, None, 'up')
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv2d(x, wt, b, None, 'up')
                e1.record()
            torch.cuda.synchronize()
            res[q].append(e0.elapsed_time(e1) / 10)
            # data gradient alone (the layer's backward with a frozen weight: dX only)
            for _ in range(2):
                ops.up_conv_subpix_dgrad(dy, wt, hw, hw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.up_conv_subpix_dgrad(dy, wt, hw, hw)
            e1.record()
            torch.cuda.synchronize()
            resg[q].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * n * 4 * hw * hw * c * c * 9
    m0, m1 = min(res[0]), min(res[1])
    print(f'{c:4d} -> {c:4d} @ {hw}^2 -> {2 * hw}^2: 8-wave sub-pixel {m0:.3f} ms ({fl / m0 / 1e9:.0f} TF/s alg) | q16up + records {m1:.3f} ms '
          f'({fl / m1 / 1e9:.0f} TF/s alg, {fl * 4 / 9 / m1 / 1e9:.0f} actual)  x{m0 / m1:.2f}   (incl. weight expand + pack each call)')
    g0, g1 = min(resg[0]), min(resg[1])
    print(f'     data gradient: 8-wave gather mode {g0:.3f} ms ({fl / g0 / 1e9:.0f} TF/s alg) | q16upg + records of dY {g1:.3f} ms '
          f'({fl / g1 / 1e9:.0f} TF/s alg, {fl * 4 / 9 / g1 / 1e9:.0f} actual)  x{g0 / g1:.2f}')
    del x, wt, dy
    torch.cuda.empty_cache()
