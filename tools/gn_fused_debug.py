import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import lib as L
lib = L.load()
n, c, G, hw = 2, 64, 32, 4096
torch.manual_seed(0)
x = torch.randn(n, c, 64, 64, device='cuda') * 2 + 0.7
dy = torch.randn_like(x)
g = 1 + 0.2 * torch.randn(c, device='cuda'); b = 0.2 * torch.randn(c, device='cuda')
stats = torch.empty(n * G * 2, device='cuda'); y = torch.empty_like(x)
ws = torch.zeros(lib.eovae_gn_workspace_bytes(n, c, hw, G), dtype=torch.uint8, device='cuda')
L.check(lib.eovae_gn_silu_fwd(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(stats), None, n, c, hw, G, 1e-6, 1, L.ptr(ws), L.stream()), 'f')
out = {}
for on in (1, 0):
    lib.eovae_gn_bwd_fused(on)
    dx = torch.empty_like(x); dg = torch.empty_like(g); db = torch.empty_like(b)
    L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(stats), None, L.ptr(dx), L.ptr(dg), L.ptr(db), None, n, c, hw, G, 1, L.ptr(ws), L.stream()), 'b')
    torch.cuda.synchronize()
    out[on] = (dx, dg, db)
d = (out[1][0] - out[0][0])
print('dx err per (n, c):', d.abs().amax(dim=(2, 3)))
print('dgamma err', (out[1][1] - out[0][1]).abs().max().item(), 'dbeta err', (out[1][2] - out[0][2]).abs().max().item())
e = d[0, 0].flatten()
print('first channel: err at 0..8', e[:8], 'at 2048..', e[2048:2056])
# is the error of the form  a + b * xhat ?
mean, rstd = stats[0].item(), stats[1].item()
xh = (x[0, 0].flatten() - mean) * rstd
A = torch.stack([torch.ones_like(xh), xh], 1)
sol = torch.linalg.lstsq(A, e[:, None]).solution.flatten()
print('fit a + b xhat:', sol, 'residual', (A @ sol - e).abs().max().item())
# granules of the fused call (run it again last so that ws holds them)
lib.eovae_gn_bwd_fused(1)
dx = torch.empty_like(x); dg = torch.empty_like(g); db = torch.empty_like(b)
L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(stats), None, L.ptr(dx), L.ptr(dg), L.ptr(db), None, n, c, hw, G, 1, L.ptr(ws), L.stream()), 'b')
torch.cuda.synchronize()
gran = ws[:n * c * (hw // 2048) * 16].view(torch.float32).view(-1, 4)
st = stats.view(-1, 2)
cpg = c // G
xs = x.view(n * G, cpg, hw).double()
xh = (xs - st[:, 0].double()[:, None, None]) * st[:, 1].double()[:, None, None]
gam = g.view(1, G, cpg, 1).expand(n, G, cpg, 1).reshape(n * G, cpg, 1).double(); bet = b.view(1, G, cpg, 1).expand(n, G, cpg, 1).reshape(n * G, cpg, 1).double()
z = xh * gam + bet
sg = torch.sigmoid(z)
dz = dy.view(n * G, cpg, hw).double() * sg * (1 + z * (1 - sg))
s1 = dz.view(n * G, cpg * hw // 2048, 2048).sum(-1)
s2 = (dz * xh).view(n * G, cpg * hw // 2048, 2048).sum(-1)
print('granule s1 (kernel / torch):', gran[:4, 0], s1.flatten()[:4].float())
print('granule s2 (kernel / torch):', gran[:4, 1], s2.flatten()[:4].float())
print('token words', [hex(v & 0xffffffff) for v in gran[:2, 2:].view(torch.int32).flatten().tolist()])
dgt = s2.view(n, G, cpg, hw // 2048).sum(-1).sum(0).flatten().float()
dbt = s1.view(n, G, cpg, hw // 2048).sum(-1).sum(0).flatten().float()
print('dgamma torch', dgt[:6]); print('dgamma fused', dg[:6]); print('dgamma 2pass', out[0][1][:6])
print('dbeta torch', dbt[:6]); print('dbeta fused', db[:6]); print('dbeta 2pass', out[0][2][:6])
