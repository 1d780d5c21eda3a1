"""Tap-by-tap probe of conv_f16c8_q16up.hip on exactly representable data: which tap's contribution goes wrong, and where from?"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from eo_vae_amd import lib as L, ops
n, cin, cout, h, w = 1, 64, 128, 8, 32
ops.CONV_PRECISION = 'f16c8'
ops.GN_FUSED_STATS = False
lib = L.load()
lib.eovae_set_option(b'c8_up_q16', 1)
torch.manual_seed(1)
# x[c][y][x] encodes its own position: c * 4096 would overflow the exact range; use small distinct integers per (y, x) and channel sign
yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
base = (yy * 32 + xx + 1).float()                       # 1 .. 256
x = torch.stack([base + 300 * (c % 7) for c in range(cin)])[None].cuda()   # exact in f16 up to 2048
for ky in range(3):
    for kx in range(3):
        wt = torch.zeros(cout, cin, 3, 3)
        for co in range(cout):
            wt[co, co % cin, ky, kx] = 1.0
        wt = wt.cuda()
        with torch.no_grad():
            y = ops.conv2d(x, wt, None, None, 'up')
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), wt, padding=1)
        torch.cuda.synchronize()
        bad = (y - ref).abs() > 1e-3
        print(f'tap ({ky},{kx}): bad {int(bad.sum())}', end='')
        if bad.any():
            t = bad.nonzero()[0].tolist()
            _, c, r, col = t
            got, want = y[0, c, r, col].item(), ref[0, c, r, col].item()
            off = 300 * ((c % cin) % 7)
            gp, wp_ = got - off - 1, want - off - 1
            print(f'  first {t}: got {got} (pixel y {int(gp) // 32} x {int(gp) % 32}) want {want} (pixel y {int(wp_) // 32} x {int(wp_) % 32})'
                  f'  bad cols {sorted(set(bad.nonzero()[:, 3].tolist()))[:12]} bad ch%16 {sorted(set((bad.nonzero()[:, 1] % 16).tolist()))}'
                  f' bad ch//16 {sorted(set((bad.nonzero()[:, 1] // 16).tolist()))}', end='')
        print()
