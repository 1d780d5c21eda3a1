"""Same-box A/B of the stride-2 layers' data gradient: through conv_f16c8_q16up.hip on MX records of dY (incl. the records pass)
against the 8-wave kernel's pixel-shuffle mode, at the encoder's three Downsample layers (B = 16), alternating."""
import sys
import torch
sys.path.insert(0, '.')
from eo_vae_amd import lib as L, ops
ops.CONV_PRECISION = 'f16c8'
for n, c, hw in [(16, 128, 256), (16, 256, 128), (16, 512, 64)]:
    torch.manual_seed(hw)
    wt = torch.randn(c, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    dy = torch.randn(n, c, hw // 2, hw // 2, device='cuda') * 1e-4
    res = {0: [], 1: []}
    for rep in range(6):
        for q in (0, 1):
            ops.DOWN_DGRAD_UP_MX = bool(q)
            for _ in range(2):
                ops.down_conv_s2d_dgrad(dy, wt, hw, hw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.down_conv_s2d_dgrad(dy, wt, hw, hw)
            e1.record()
            torch.cuda.synchronize()
            res[q].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * n * (hw // 2) ** 2 * c * c * 9
    m0, m1 = min(res[0]), min(res[1])
    print(f'{c:4d} -> {c:4d} @ {hw}^2 -> {hw // 2}^2, data gradient: 8-wave pixel-shuffle mode {m0:.3f} ms ({fl / m0 / 1e9:.0f} TF/s alg) | '
          f'q16up on records of dY {m1:.3f} ms ({fl / m1 / 1e9:.0f} TF/s alg)  x{m0 / m1:.2f}   (incl. weight expand + pack each call)')
