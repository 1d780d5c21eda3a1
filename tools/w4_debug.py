import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops
ops.CONV_PRECISION = 'f16c8'
n, cin, cout, h, w = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (5, 32, 256, 128, 128)
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g).cuda()
wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
def run(v, ni):
    from eo_vae_amd import lib as _L
    _L.load().eovae_set_option(b'c8_w4', int(v))
    _L.load().eovae_set_option(b'c8_w4_ni', int(ni))
    with torch.no_grad():
        y = ops.conv2d(x, wt, None, None, 'normal')
    torch.cuda.synchronize()
    return y
y0 = run('0', '4')
for ni in ('4', '2'):
    y1 = run('1', ni)
    th = 4 * int(ni)
    d = (y1 - y0).abs().view(n, cout // 128, 128, h // th, th, w // 32, 32).amax(dim=(2, 4, 6))   # [n][cob][ty][tx]
    bad = (d > 1e-3).nonzero()
    print('NI', ni, 'bad tiles:', len(bad), 'of', d.numel())
    for b in bad[:40].tolist():
        print('   n cob ty tx =', b, 'err', float(d[tuple(b)]))
    if len(bad):
        nb, cb, ty, tx = bad[0].tolist()
        e = (y1 - y0).abs()[nb, cb * 128:(cb + 1) * 128, ty * th:(ty + 1) * th, tx * 32:(tx + 1) * 32]
        print('   first bad tile: per row max', e.amax(dim=(0, 2)).tolist())
        print('   per 32-channel co-tile max', e.view(4, 32, th, 32).amax(dim=(1, 2, 3)).tolist())
