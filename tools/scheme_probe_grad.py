"""CPU emulation of the split-operand schemes on the TRAIN step of the tiny model: forward, data
gradient and weight gradient of every 3x3 convolution with Cin % 32 == 0 run through the scheme;
loss and every parameter gradient are compared with the fp32 oracle (bars: loss 1e-4, gradients
5e-4 max-rel per tensor, tests/test_gpu_model.py).  Companion of tools/scheme_probe_cpu.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import eo_vae_cpu as O
from tools.scheme_probe_cpu import bf16, e4m3, e5m2, f16, mx_quant, pow2_scale

# operand kinds: 'act' = activation or gradient tensor (no amax known unless scaled), 'w' = weights


def split_bf16x3(a, b, op, ka, kb):
    ah, bh = bf16(a), bf16(b)
    return op(ah, bh) + op(ah, bf16(b - bh)) + op(bf16(a - ah), bh)


def make_f16c8(scale_acts):
    def split(a, b, op, ka, kb):
        def prep(t, kind):
            # power-of-two per-tensor scale: weights always (pack time); activations / gradients when asked
            s = pow2_scale(t, 2.0 ** 14) if (kind == 'w' or scale_acts) else 1.0
            ts = t * s
            th = f16(ts)
            tl = ts - th
            if kind == 'w':   # e4m3 images, own power-of-two scales
                s8h, s8l = pow2_scale(ts, 256.0), (pow2_scale(tl, 256.0) if tl.abs().max() > 0 else 1.0)
                return s, th, e4m3(ts * s8h) / s8h, e4m3(tl * s8l) / s8l
            return s, th, e5m2(ts), e5m2(tl * 4096.0) / 4096.0
        sa, ah, a8h, a8l = prep(a, ka)
        sb, bh, b8h, b8l = prep(b, kb)
        return (op(ah, bh) + (op(a8l, b8h) + op(a8h, b8l))) / (sa * sb)
    return split


def make_mx(fmt, where=('fwd', 'dgrad', 'wgrad')):
    """Round 4: f16 main product + the correction product on block-scaled `fmt` operands (per-32 E8M0 scales along the
    K axis of the product: channels forward / data gradient, pixels of a row in the weight gradient) = 1.5 units.
    `where`: the products that carry the correction at all (the others run plain f16: 1 unit)."""
    def split(a, b, op, ka, kb):
        kind = CUR['op']
        da, db = CUR['kdims']
        sa = pow2_scale(a, 2.0 ** 14) if ka == 'act' else pow2_scale(a, 2.0 ** 14)
        sb = pow2_scale(b, 2.0 ** 14)
        a_, b_ = a * sa, b * sb
        ah, bh = f16(a_), f16(b_)
        main = op(ah, bh)
        if kind not in where:
            return main / (sa * sb)
        al, bl = a_ - ah, b_ - bh
        corr = op(mx_quant(al, da, fmt), mx_quant(b_, db, fmt)) + op(mx_quant(a_, da, fmt), mx_quant(bl, db, fmt))
        return (main + corr) / (sa * sb)
    return split


SPLITS = {'f32': lambda a, b, op, ka, kb: op(a, b), 'bf16x3': split_bf16x3,
          'f16+c8 (dy scaled)': make_f16c8(True), 'f16+c8 (no act scale)': make_f16c8(False),
          'f16+mx e2m3 (1.5 u)': make_mx('e2m3'), 'f16+mx e3m2 (1.5 u)': make_mx('e3m2'), 'f16+mx e2m1 (1.5 u)': make_mx('e2m1'),
          'f16+mx e2m3, dgrad plain f16 (1.33 u)': make_mx('e2m3', ('fwd', 'wgrad')),
          'f16+c8 fwd/wgrad, dgrad plain f16 (1.67 u)': None}
CUR = {'fn': None, 'op': 'fwd', 'kdims': (1, 1)}
_C8 = make_f16c8(True)
SPLITS['f16+c8 fwd/wgrad, dgrad plain f16 (1.67 u)'] = (
    lambda a, b, op, ka, kb: (op(f16(a * pow2_scale(a, 2.0 ** 14)), f16(b * pow2_scale(b, 2.0 ** 14))) / (pow2_scale(a, 2.0 ** 14) * pow2_scale(b, 2.0 ** 14)))
    if CUR['op'] == 'dgrad' else _C8(a, b, op, ka, kb))
_orig = F.conv2d


class EmuConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, padding)
        CUR['op'], CUR['kdims'] = 'fwd', (1, 1)
        return CUR['fn'](x, w, lambda a, b: _orig(a, b, None, stride, padding), 'act', 'w')

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding = ctx.sp
        CUR['op'], CUR['kdims'] = 'dgrad', (1, 0)
        dx = CUR['fn'](dy, w, lambda a, b: torch.nn.grad.conv2d_input(x.shape, b, a, stride, padding), 'act', 'w')
        CUR['op'], CUR['kdims'] = 'wgrad', (3, 3)
        dw = CUR['fn'](x, dy, lambda a, b: torch.nn.grad.conv2d_weight(a, w.shape, b, stride, padding), 'act', 'act')
        return dx, dw, None, None


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if CUR['fn'] is not None and w.shape[-1] == 3 and w.shape[1] % 32 == 0 and groups == 1:
        y = EmuConv.apply(x, w, stride, padding)
        return y + b.view(1, -1, 1, 1) if b is not None else y
    return _orig(x, w, b, stride, padding, dilation, groups)


def run(name, gscale):
    CUR['fn'] = None if name == 'f32' else SPLITS[name]
    F.conv2d = conv2d
    torch.nn.functional.conv2d = conv2d
    try:
        cfg = O.model_cfg(**O.TINY)
        m = O.build_vae(cfg)
        O.det_fill(m, seed=21)
        m.train()
        wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
        x = O.synthetic_batch(2, 3, 64, seed=7)
        eps = torch.randn(2, 32, 8, 8, generator=torch.Generator().manual_seed(8))
        rec, post = m(x, wvs, noise=eps)
        loss = O.charbonnier(rec, x) + 1e-6 * post.kl().mean()
        (loss * gscale).backward()   # gscale shrinks every dy: exercises the fp16 underflow range
        return loss.item(), {k: p.grad.clone() / gscale for k, p in m.named_parameters() if p.grad is not None}
    finally:
        F.conv2d = _orig
        torch.nn.functional.conv2d = _orig


if __name__ == '__main__':
    torch.set_num_threads(8)
    for gscale in (1.0, 2.0 ** -12):
        l0, g0 = run('f32', gscale)
        print(f'--- upstream gradient scale {gscale:g}')
        for name in list(SPLITS)[1:]:
            l, g = run(name, gscale)
            tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g0.values())))
            worst = max(((g[k] - g0[k]).double().norm().item() / max(g0[k].double().norm().item(), 1e-3 * tot))
                        for k in g0)   # the metric of tests/test_gpu_model.py (bar 5e-4)
            med = float(np.median([((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item() for k in g0]))
            print(f'{name:24s} loss rel {abs(l - l0) / abs(l0):.2e}  grad worst (test metric, bar 5e-4) {worst:.2e}  '
                  f'rms-rel median {med:.2e}', flush=True)
