import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import eo_vae_cpu as O
from tests.test_gpu_model import build_pair, T, rel
g = dict(np.load('/root/repo/tests/golden/tiny_train.npz'))
oracle, prod = build_pair(O.TINY, 21)
oracle.train(); prod.train()
wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
x, eps = T(g['x0']), T(g['eps0'])
acts_o, acts_p = {}, {}
def hook(store, name):
    def f(m, i, o):
        if isinstance(o, torch.Tensor):
            o.retain_grad() if o.requires_grad else None
            store[name] = o
    return f
for n_, m in oracle.named_modules():
    if n_ and n_.count('.') <= 3: m.register_forward_hook(hook(acts_o, n_))
for n_, m in prod.named_modules():
    if n_ and n_.count('.') <= 3: m.register_forward_hook(hook(acts_p, n_))
rec_o, post_o = oracle(x, wvs, noise=eps)
loss_o = torch.nn.functional.l1_loss(rec_o, x); loss_o.backward()
from eo_vae_amd import ops
rec_p, post_p = prod(x.cuda(), wvs.cuda(), noise=eps.cuda())
loss_p = ops.l1_loss(rec_p, x.cuda()); loss_p.backward()
print('loss', loss_o.item(), loss_p.item())
for k in acts_o:
    if k in acts_p and acts_o[k].shape == acts_p[k].shape:
        a, b = acts_p[k], acts_o[k]
        ge = rel(a.grad, b.grad) if (a.grad is not None and b.grad is not None) else float('nan')
        print('%-40s fwd %.2e  grad %.2e' % (k, rel(a, b), ge))
