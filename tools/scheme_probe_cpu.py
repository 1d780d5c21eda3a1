"""CPU emulation of candidate split-operand arithmetic schemes for the 3x3 convolutions
(VERDICT r1 item 2): every F.conv2d of the ORACLE with a 3x3 kernel and Cin % 32 == 0 is replaced
by the scheme's sum of low-precision products (fp32 accumulation, as the MFMA does), the whole
model is run on the golden cfg1 input and the error vs the reference's fp32 output is reported.
Forward only; tools/scheme_probe_grad.py covers the gradients on the tiny model.

  units = matrix-core cycles per algorithmic product in units of one bf16/f16 MFMA
          (the f8f6f4 block-scaled MFMA runs 2x the bf16 rate, MI355X_MICROARCH.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import eo_vae_cpu as O

bf16 = lambda t: t.to(torch.bfloat16).float()
f16 = lambda t: t.to(torch.float16).float()
e5m2 = lambda t: t.clamp(-57344, 57344).to(torch.float8_e5m2).float()
e4m3 = lambda t: t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def pow2_scale(t, target):
    """power of two s with amax(t) * s in (target/2, target]"""
    a = t.abs().max().item()
    return 2.0 ** np.floor(np.log2(target / a)) if a > 0 else 1.0


SCHEMES = {}


def scheme(name, units):
    def deco(fn):
        SCHEMES[name] = (fn, units)
        return fn
    return deco


@scheme('f32', 16)
def _f32(x, w, conv):
    return conv(x, w)


@scheme('bf16x3', 3)
def _bf16x3(x, w, conv):
    xh, wh = bf16(x), bf16(w)
    xl, wl = bf16(x - xh), bf16(w - wh)
    return conv(xh, wh) + conv(xh, wl) + conv(xl, wh)


@scheme('bf16', 1)
def _bf16(x, w, conv):
    return conv(bf16(x), bf16(w))


@scheme('f16', 1)
def _f16(x, w, conv):
    return conv(f16(x), f16(w))


@scheme('f16x2_whi', 2)   # VERDICT (i): f16 hi/lo activations x f16-hi weights
def _f16x2(x, w, conv):
    xh, wh = f16(x), f16(w)
    return conv(xh, wh) + conv(f16(x - xh), wh)


@scheme('bf16x2_whi', 2)  # VERDICT (ii): drop hi_x * lo_w
def _bf16x2(x, w, conv):
    xh, wh = bf16(x), bf16(w)
    return conv(xh, wh) + conv(bf16(x - xh), wh)


def _corr8(x, w, conv, hi, act8, w8, sx_lo=2.0 ** 12):
    """hi*hi in 16-bit + both correction terms as ONE K-concatenated fp8 product (1 unit)."""
    xh, wh = hi(x), hi(w)
    xl, wl = x - xh, w - wh
    sw = pow2_scale(w, 256.0)            # weights: per-tensor amax known at pack time
    swl = pow2_scale(wl, 256.0) if wl.abs().max() > 0 else 1.0
    if act8 is e4m3:                     # activations in e4m3 need a per-tensor scale too
        sxh, sxl = pow2_scale(x, 256.0), (pow2_scale(xl, 256.0) if xl.abs().max() > 0 else 1.0)
    else:                                # e5m2 has fp16's exponent range: fixed scales
        sxh, sxl = 1.0, sx_lo
    c1 = conv(act8(xl * sxl), w8(w * sw)) / (sxl * sw)
    c2 = conv(act8(x * sxh), w8(wl * swl)) / (sxh * swl)
    return conv(xh, wh) + (c1 + c2)


@scheme('f16+e5m2/e4m3', 2)
def _f16_c8a(x, w, conv):
    return _corr8(x, w, conv, f16, e5m2, e4m3)


@scheme('f16+e4m3/e4m3', 2)
def _f16_c8b(x, w, conv):
    return _corr8(x, w, conv, f16, e4m3, e4m3)


@scheme('bf16+e5m2/e4m3', 2)
def _bf16_c8(x, w, conv):
    return _corr8(x, w, conv, bf16, e5m2, e4m3, sx_lo=2.0 ** 9)


# ---- round 4: 1.5-unit candidates -- the correction product on the block-scaled FP6 / FP4 rate.
# v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 / e3m2 / e2m1 operands runs at 4x the f16 rate (MI355X_MICROARCH.md:45):
# per 32 channels 2 f16 MFMAs (64 cycles) + ONE K = 64 product [xl | xh] . [wh | wl] in 32 cycles = 1.5 units.  The
# format is built for per-32-element E8M0 scales (OCP MX): emulated here with the shared exponent floor(log2(amax of the
# 32 K-neighbours)) - emax(format), round-to-nearest onto the format's grid, saturation at its maximum.
def _grid(ebits, mbits):
    bias = (1 << (ebits - 1)) - 1
    vals = [m / (1 << mbits) * 2.0 ** (1 - bias) for m in range(1 << mbits)]                       # subnormals (e = 0)
    vals += [(1 + m / (1 << mbits)) * 2.0 ** (e - bias) for e in range(1, 1 << ebits) for m in range(1 << mbits)]
    return torch.tensor(sorted(set(vals)), dtype=torch.float32)


FORMATS = {'e2m3': _grid(2, 3), 'e3m2': _grid(3, 2), 'e2m1': _grid(2, 1)}     # fp6, fp6, fp4 (no inf / nan codes)


def mx_quant(t, dim, fmt, block=32):
    """MX-style block quantisation of `t` along `dim` (blocks of 32 neighbours share a power-of-two scale)."""
    grid = FORMATS[fmt]
    emax = int(np.floor(np.log2(grid[-1].item())))
    t = t.movedim(dim, -1)
    shp = t.shape
    pad = (-shp[-1]) % block
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(*t.shape[:-1], -1, block)
    amax = b.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(1e-38))) - emax
    scale = torch.exp2(e)
    y = (b / scale).clamp(-grid[-1].item(), grid[-1].item())
    idx = torch.bucketize(y.abs(), (grid[1:] + grid[:-1]) / 2)          # nearest grid point
    q = grid[idx] * torch.sign(y) * scale
    q = torch.where(amax > 0, q, torch.zeros_like(q)).reshape(*t.shape)
    if pad:
        q = q[..., :shp[-1]]
    return q.movedim(-1, dim)


def _corr_mx(x, w, conv, fmt_act, fmt_w):
    """f16 main product + both correction terms on block-scaled `fmt` operands (scales per 32 channels: per pixel for the
    activations, per (co, tap) for the weights -- the K axis of the forward convolution)."""
    xh, wh = f16(x), f16(w)
    xl, wl = x - xh, w - wh
    c1 = conv(mx_quant(xl, 1, fmt_act), mx_quant(w, 1, fmt_w))
    c2 = conv(mx_quant(x, 1, fmt_act), mx_quant(wl, 1, fmt_w))
    return conv(xh, wh) + (c1 + c2)


@scheme('f16+mx e2m3/e2m3', 1.5)
def _f16_mx_a(x, w, conv):
    return _corr_mx(x, w, conv, 'e2m3', 'e2m3')


@scheme('f16+mx e3m2/e2m3', 1.5)
def _f16_mx_b(x, w, conv):
    return _corr_mx(x, w, conv, 'e3m2', 'e2m3')


@scheme('f16+mx e3m2/e3m2', 1.5)
def _f16_mx_c(x, w, conv):
    return _corr_mx(x, w, conv, 'e3m2', 'e3m2')


@scheme('f16+mx e2m1/e2m1', 1.5)   # FP4 operands (same rate as FP6: only the bytes differ)
def _f16_mx_d(x, w, conv):
    return _corr_mx(x, w, conv, 'e2m1', 'e2m1')


@scheme('f16+mx fp8 blocks', 2)    # control: e4m3 with the same per-32 block scales (what the scales alone buy over f16c8)
def _f16_mx_e(x, w, conv):
    FORMATS.setdefault('e4m3g', _grid(4, 3)[:-1])
    return _corr_mx(x, w, conv, 'e4m3g', 'e4m3g')


class Patch:
    def __init__(self, fn):
        self.fn, self.orig = fn, F.conv2d

    def __enter__(self):
        orig, fn = self.orig, self.fn

        def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            if w.shape[-1] == 3 and w.shape[1] % 32 == 0 and groups == 1:
                conv = lambda a, c: orig(a, c, None, stride, padding, dilation, groups)
                y = fn(x, w, conv)
                return y + b.view(1, -1, 1, 1) if b is not None else y
            return orig(x, w, b, stride, padding, dilation, groups)
        F.conv2d = conv2d
        torch.nn.functional.conv2d = conv2d
        return self

    def __exit__(self, *a):
        F.conv2d = self.orig
        torch.nn.functional.conv2d = self.orig


def main(names):
    torch.set_num_threads(8)
    g = dict(np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'full_cfg1.npz')))
    oracle = O.build_vae(O.model_cfg())
    O.det_fill(oracle, seed=31)
    oracle.eval()
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    x = O.synthetic_batch(2, 3, 256, seed=1234)
    r0, s0 = torch.from_numpy(g['recon_sub']).double(), torch.from_numpy(g['spatial_sub']).double()
    print(f'{"scheme":18s} units  recon max-rel  rms-rel | latent max-rel  rms-rel')
    for name in names:
        fn, units = SCHEMES[name]
        with torch.no_grad(), Patch(fn):
            rec = oracle.reconstruct(x, wvs)[:, :, ::8, ::8].double()
            spat = oracle.encode_spatial_normalized(x, wvs)[:, :, ::2, ::2].double()
        print(f'{name:18s} {units:5g}  {((rec - r0).abs().max() / r0.abs().max()).item():.2e}  '
              f'{((rec - r0).norm() / r0.norm()).item():.2e} | '
              f'{((spat - s0).abs().max() / s0.abs().max()).item():.2e}  '
              f'{((spat - s0).norm() / s0.norm()).item():.2e}', flush=True)


if __name__ == '__main__':
    main(sys.argv[1:] or list(SCHEMES))
