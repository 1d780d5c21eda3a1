"""Where does conv_f16c8_q16up.hip differ from the 8-wave sub-pixel kernel?  python tools/q16up_debug.py n cin cout h w"""
import sys
import torch
sys.path.insert(0, '.')
from eo_vae_amd import lib as L, ops
n, cin, cout, h, w = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (2, 64, 128, 16, 32)
ops.CONV_PRECISION = 'f16c8'
lib = L.load()
torch.manual_seed(cin + h)
x = (torch.randn(n, cin, h, w) * 3 + 0.5).cuda()
wt = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda()
b = torch.randn(cout).cuda()
ys = []
for q in (1, 0):
    lib.eovae_set_option(b'c8_up_q16', q)
    with torch.no_grad():
        ys.append(ops.conv2d(x, wt, b, None, 'up'))
torch.cuda.synchronize()
e = (ys[0] - ys[1]).abs() / ys[1].abs().max()
print('max rel err', e.max().item())
bad = e > 1e-4
print('bad fraction', bad.float().mean().item())
for py in range(2):
    for px in range(2):
        print(f'parity ({py},{px}): max {e[:, :, py::2, px::2].max().item():.3e} bad {bad[:, :, py::2, px::2].float().mean().item():.4f}')
print('per image  ', [f'{v:.2e}' for v in e.amax((1, 2, 3)).tolist()])
print('per hi row ', [f'{v:.1e}' for v in e.amax((0, 1, 3)).tolist()])
print('per hi col ', [f'{v:.1e}' for v in e.amax((0, 1, 2)).tolist()])
print('per 16 ch  ', [f'{v:.1e}' for v in e.reshape(n, cout // 16, 16, 2 * h, 2 * w).amax((0, 2, 3, 4)).tolist()])
print('per ch%16  ', [f'{v:.1e}' for v in e.reshape(n, cout // 16, 16, 2 * h, 2 * w).amax((0, 1, 3, 4)).tolist()])
idx = bad.nonzero()
print('bad count', idx.shape[0])
yn, yo = ys[0], ys[1]
for t in idx[:12].tolist():
    nn, c, r, col = t
    print(t, 'new', yn[nn, c, r, col].item(), 'old', yo[nn, c, r, col].item(), 'bias', b[c].item(),
          'old neighbours', [round(v, 4) for v in yo[nn, c, r, max(0, col - 3):col + 4].tolist()],
          'new neighbours', [round(v, 4) for v in yn[nn, c, r, max(0, col - 3):col + 4].tolist()])
# does the bad value equal the good value somewhere else?
t = idx[0].tolist()
v = yn[t[0], t[1], t[2], t[3]]
hit = ((yo - v).abs() < 1e-5 * yo.abs().max()).nonzero()
print('same value in old at', hit[:8].tolist())
# difference new - old for the bad entries of one (image, row): a pattern over channels?
d = (yn - yo)[0, :, 0, 5]
print('diff at image 0 row 0 col 5 per channel', [round(x_, 3) for x_ in d.tolist()])
print('--- bad co tiles per (image, low-res row, py) [px = 1]')
for nn in range(n):
    for ly in range(h):
        for py in range(2):
            bb = bad[nn, :, 2 * ly + py, 1::2].reshape(cout // 16, 16, w).any(2)   # [i][li]
            tiles = [(i, [l for l in range(16) if bb[i, l]]) for i in range(cout // 16) if bb[i].any()]
            cols = bad[nn, :, 2 * ly + py, :].any(0).nonzero().flatten().tolist()
            print(nn, ly, py, tiles, cols)
