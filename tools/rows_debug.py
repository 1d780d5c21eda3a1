"""Debug helper: where does the row-streaming weight gradient differ from the tile kernel?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops, lib as L
lib = L.load()
ops.CONV_PRECISION = 'f16c8'
B, cin, cout, h, w = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 64, 128, 16, 32))]
torch.manual_seed(0)
x = torch.randn(B, cin, h, w, device='cuda')
dy = torch.randn(B, cout, h, w, device='cuda') * 1e-3
dw = torch.zeros(cout, cin, 3, 3, device='cuda')
db = torch.zeros(cout, device='cuda')
amax = ops.absmax_bits(dy)
xrec = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True).detach()
ws = torch.empty(lib.eovae_conv3x3_f16c8_wgrad_dw_workspace(cin, cout) // 4, device='cuda')
res = {}
for v in (0, 1, 1):
    lib.eovae_set_option(b'c8_wgrad_rows', v)
    dw.zero_(); db.zero_()
    L.check(lib.eovae_conv3x3_f16c8_wgrad_dw(L.ptr(xrec), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(amax), None, B, cin, h, w, cout, h, w, 1, L.ptr(ws), L.stream()), 'wgrad')
    torch.cuda.synchronize()
    res.setdefault(v, []).append((dw.clone(), db.clone()))
t, r1, r2 = res[0][0][0], res[1][0][0], res[1][1][0]
print('rows run1 == run2:', torch.equal(r1, r2), ' nan in rows:', int(torch.isnan(r1).sum()), 'of', r1.numel())
sc = t.abs().max().item()
for ky in range(3):
    for kx in range(3):
        d = (r1[:, :, ky, kx] - t[:, :, ky, kx])
        print(f'tap ({ky},{kx}): max rel diff {torch.nan_to_num(d, nan=9e9).abs().max().item() / sc:.3e}  nan {int(torch.isnan(d).sum())}')
d = torch.nan_to_num(r1 - t, nan=9e9).abs().amax((2, 3)) / sc
print('per 32x32 (co-tile, ci-tile) block max rel diff:')
for a in range(cout // 32):
    print(' '.join(f'{d[a*32:(a+1)*32, b*32:(b+1)*32].max().item():.1e}' for b in range(cin // 32)))
print('db rel diff', ((res[1][0][1] - res[0][0][1]).abs().max() / res[0][0][1].abs().max()).item())
