"""CPU emulation (round 5): which products of the train step need the correction product at all?  The weight gradient is a LEAF
of the computation -- its rounding errors feed no further product -- so it is tried as ONE f16 x f16 product while the
forward and the data gradient keep theirs (f16c8, or the MX-FP6 form).  Same tiny-model train step and metric as
tools/scheme_probe_grad.py (worst tensor: ||g - g0|| / max(||g0||, 1e-3 ||g_all||), bar 1e-3 for the f16c8 mode).
    python tools/scheme_probe_wgrad.py        -> profiles/r05_scheme_probe_wgrad.log"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import tools.scheme_probe_grad as G
from tools.scheme_probe_cpu import f16, pow2_scale

def plain_f16(a, b, op, ka, kb):
    sa, sb = pow2_scale(a, 2.0 ** 14), pow2_scale(b, 2.0 ** 14)
    return op(f16(a * sa), f16(b * sb)) / (sa * sb)

def mixed(base, plain_ops):
    def split(a, b, op, ka, kb):
        if G.CUR['op'] in plain_ops:
            return plain_f16(a, b, op, ka, kb)
        return base(a, b, op, ka, kb)
    return split

def one_sided(base):
    # wgrad: f16 x f16 + (dy_lo x X) only  [a = x, b = dy in the wgrad call]
    def split(a, b, op, ka, kb):
        if G.CUR['op'] != 'wgrad':
            return base(a, b, op, ka, kb)
        sa, sb = pow2_scale(a, 2.0 ** 14), pow2_scale(b, 2.0 ** 14)
        a_, b_ = a * sa, b * sb
        ah, bh = f16(a_), f16(b_)
        from tools.scheme_probe_cpu import mx_quant
        corr = op(mx_quant(a_, 3, 'e2m3'), mx_quant(b_ - bh, 3, 'e2m3'))
        return (op(ah, bh) + corr) / (sa * sb)
    return split

C8 = G.make_f16c8(True)
MX = G.make_mx('e2m3')
G.SPLITS.clear()
G.SPLITS.update({'f32': None,
  'f16c8 all (today)': C8,
  'stage 1: mx fwd, c8 dgrad/wgrad': (lambda a, b, op, ka, kb: (MX if G.CUR['op'] == 'fwd' else C8)(a, b, op, ka, kb)),
  'c8 fwd/dgrad, wgrad PLAIN f16': mixed(C8, ('wgrad',)),
  'mx fwd/dgrad, wgrad PLAIN f16': mixed(MX, ('wgrad',)),
  'mx fwd + c8 dgrad, wgrad PLAIN f16': mixed((lambda a, b, op, ka, kb: (MX if G.CUR['op'] == 'fwd' else C8)(a, b, op, ka, kb)), ('wgrad',)),
  'mx fwd/dgrad, wgrad f16 + dy_lo.X only': one_sided(MX),
})
torch.set_num_threads(8)
for gscale in (1.0,):
    l0, g0 = G.run('f32', gscale)
    for name in list(G.SPLITS)[1:]:
        l, g = G.run(name, gscale)
        tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g0.values())))
        per = {k: ((g[k] - g0[k]).double().norm().item() / max(g0[k].double().norm().item(), 1e-3 * tot)) for k in g0}
        worst = max(per.values()); wk = max(per, key=per.get)
        # entry-wise metric of oracle/parity_checks.py: max |d| / max(rms, max|ref sample|) approximated over ALL entries vs tensor rms
        ent = max(((g[k] - g0[k]).abs().max().item() / max(g0[k].double().norm().item() / np.sqrt(g0[k].numel()), 1e-3 * tot / np.sqrt(g0[k].numel()))) for k in g0)
        med = float(np.median([((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item() for k in g0]))
        conv_w = [k for k in g0 if g0[k].dim() == 4 and g0[k].shape[-1] == 3 and g0[k].shape[1] % 32 == 0]
        wmed = float(np.median([per[k] for k in conv_w])); wmax = max(per[k] for k in conv_w)
        print(f'{name:44s} loss {abs(l - l0) / abs(l0):.1e} | worst tensor {worst:.2e} ({wk[-40:]}) | median {med:.2e} | 3x3 weights: median {wmed:.2e} max {wmax:.2e} | worst entry / rms {ent:.2e}', flush=True)
