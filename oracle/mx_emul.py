"""TEST INFRASTRUCTURE (only tests/, bench.py's checker legs and __graft_entry__.smoke() may import this).

CPU restatement of the MX-FP6 operand arithmetic of the HIP kernels (csrc/conv_f16c8_common.h, round 5): what the
block-scaled e2m3 operands of the correction product hold, so that a kernel result can be checked against the SAME
products in float64 (agreement ~1e-6: summation order only) and not just against fp32 within the scheme's error.
Not a restatement of anything in the reference -- the reference computes in fp32; the parity of the scheme as a whole
is what tests/test_gpu_model.py pins against the reference's fixtures.

    block   = 32 neighbours along the product's K axis
    scale   = 2^(biased exponent of the block amax - 2 - 127), byte clamped to >= 1   (OCP MX: amax lands in [4, 8))
    element = e2m3: 0, 0.125 .. 0.875 (subnormal), 1 .. 1.875, 2 .. 3.75, 4 .. 7.5; round to nearest even, saturate at 7.5
    layout  = 32 x 6 bits per block, field 2i <- k = i, field 2i + 1 <- k = 16 + i (v_cvt_scalef32_2xpk16_fp6_f32), then the
              scale byte in dword 6 (tools/probe_cvt_fp6.hip, profiles/r05_probe_cvt_fp6.log)
"""
import numpy as np
import torch


def f16(t):
    return t.to(torch.float16).to(t.dtype)


def scale_bytes(amax):
    """E8M0 scale byte of blocks whose largest magnitudes are `amax` (float32 tensor)."""
    bits = amax.contiguous().float().view(torch.int32)
    return (((bits >> 23) & 0xff) - 2).clamp_min(1)


def e2m3_round(y):
    """Nearest e2m3 value of y (float64, already divided by the block scale): ties to even, saturation at 7.5."""
    a = y.abs()
    e = torch.floor(torch.log2(a.clamp_min(1.0))).clamp(0, 2)
    step = torch.exp2(e - 3)
    q = (torch.round(a / step) * step).clamp_max(7.5)      # torch.round: half to even
    return torch.sign(y) * q


def mx6_quant(t, dim):
    """Block-scaled e2m3 image of t along `dim` (length a multiple of 32), as float64 values."""
    t = t.float().movedim(dim, -1)
    shp = t.shape
    assert shp[-1] % 32 == 0
    b = t.reshape(*shp[:-1], -1, 32)
    sb = scale_bytes(b.abs().amax(-1, keepdim=True))
    scale = torch.exp2(sb.double() - 127)
    q = e2m3_round(b.double() / scale) * scale
    return q.reshape(shp).movedim(-1, dim)


def split_product(x, w, conv, kdim_x=1, kdim_w=1):
    """x * w as the f16c8 / MX kernels compute it, in float64: f16 leading parts + the two correction terms on MX-FP6
    operands.  conv(a, b): the float64 contraction; kdim_*: the K axis of each operand."""
    xh, wh = f16(x.float()), f16(w.float())
    xl, wl = x.float() - xh, w.float() - wh
    return (conv(xh.double(), wh.double())
            + conv(mx6_quant(xl, kdim_x), mx6_quant(w, kdim_w))
            + conv(mx6_quant(x, kdim_x), mx6_quant(wl, kdim_w)))


_CODE = np.array([(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) * (-1 if s else 1)
                  for s in range(2) for e in range(4) for m in range(8)])


def decode_blocks(pieces):
    """pieces: int32 array [..., 8] = a block's two 16-byte pieces {d0..d3 | d4 d5 scale pad} -> float64 [..., 32] in k order."""
    pieces = np.ascontiguousarray(pieces.astype(np.int32))
    raw = pieces[..., :6].copy().view(np.uint8).reshape(*pieces.shape[:-1], 24)
    bits = np.unpackbits(raw, axis=-1, bitorder='little').reshape(*pieces.shape[:-1], 32, 6)
    codes = (bits * (1 << np.arange(6))).sum(-1)
    vals = _CODE[codes]                                             # field order
    k_of_field = np.array([f // 2 + 16 * (f % 2) for f in range(32)])
    out = np.empty_like(vals)
    out[..., k_of_field] = vals
    scale = np.exp2((pieces[..., 6] & 0xff).astype(np.float64) - 127)
    return out * scale[..., None]
