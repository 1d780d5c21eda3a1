"""Per-parity check of conv_f16c8_q16upg.hip against the float64 emulation of its products."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from eo_vae_amd import lib as L, ops
from oracle import mx_emul as M
from test_gpu_up_mx import folded_weights, prescale_of, rel
ops.CONV_PRECISION = 'f16c8'
lib = L.load()
n, cin, cout, h, w = 2, 128, 64, 16, 32
torch.manual_seed(160)
wt = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda()
we = folded_weights(wt.cpu())
full = (torch.randn(n, cout, 2 * h, 2 * w) * 3e-4)
for par in [0, 1, 2, 3, -1]:
    dy = torch.zeros_like(full)
    if par >= 0:
        dy[:, :, (par >> 1)::2, (par & 1)::2] = full[:, :, (par >> 1)::2, (par & 1)::2]
    else:
        dy = full.clone()
    dyg = dy.cuda()
    res = {}
    for q in (1, 0):
        lib.eovae_set_option(b'c8_up_q16', q)
        res[q] = ops.up_conv_subpix_dgrad(dyg, wt, h, w)
    torch.cuda.synchronize()
    ds = prescale_of(dy)
    dyc = dy * ds
    want = torch.zeros(n, cin, h, w, dtype=torch.float64)
    exact = torch.zeros(n, cin, h, w, dtype=torch.float64)
    for p in range(4):
        plane = dyc[:, :, (p >> 1)::2, (p & 1)::2]
        wf = we[p].flip(2, 3).transpose(0, 1).contiguous()
        conv = lambda a_, b_: F.conv2d(a_, b_, padding=1)
        want += M.split_product(plane, wf, conv)
        exact += conv(plane.double(), wf.double())
    want, exact = want / ds, exact / ds
    e = (res[1].double().cpu() - want).abs() / want.abs().max()
    eimg = [f"{v:.1e}" for v in e.amax((1, 2, 3)).tolist()]
    print(eimg, end=' ')
    print(f'parity {par}: new vs emulation {rel(res[1], want):.2e}  new vs exact {rel(res[1], exact):.2e}  old vs exact {rel(res[0], exact):.2e} '
          f' emulation vs exact {rel(want, exact):.2e}; worst at {np.unravel_index(int(e.argmax()), e.shape)}')
# the two ways the data-gradient pack is made
wp_single = ops._subpix_packed(wt, True, True)
wp_pair = ops._subpix_packed_pair(wt)[1]
torch.cuda.synchronize()
print('packs equal:', torch.equal(wp_single, wp_pair), wp_single.numel(), wp_pair.numel())
if not torch.equal(wp_single, wp_pair):
    d = (wp_single != wp_pair).nonzero().flatten()
    print('differing bytes', d.numel(), 'first', d[:8].tolist(), 'tile offsets', sorted(set((d % 24576 // 8192).tolist())))
lib.eovae_set_option(b'c8_up_q16', 1)
dyg = full.cuda()
r1 = ops.up_conv_subpix_dgrad(dyg, wt, h, w, wp=wp_pair)
print('with the pair pack: new vs emulation', rel(r1, want))
# the test's flow: through autograd
lib.eovae_set_option(b'c8_up_q16', 1)
x = torch.zeros(n, cin, h, w, device='cuda', requires_grad=True)
y = ops.conv2d(x, wt, None, None, 'up')
y.backward(dyg)
torch.cuda.synchronize()
print('autograd flow: dx vs emulation', rel(x.grad, want), ' vs direct call', rel(x.grad, r1), ' y is zero:', float(y.abs().max()))
x2 = torch.zeros(n, cin, h, w, device='cuda', requires_grad=True)
with torch.no_grad():
    pass
y2 = ops.conv2d(x2, wt, None, None, 'up')
r2 = ops.up_conv_subpix_dgrad(dyg, wt, h, w)
torch.cuda.synchronize()
print('direct call after a forward: vs emulation', rel(r2, want))
