"""autograd bindings of csrc/generator.hip: the transformer encoder layer and the Linear layers of the wavelength-
conditioned weight generators (reference: modules/dynamic_conv.py:62-183, 336-366) as small exact-fp32 HIP kernels.

One encoder layer is 9 launches forward and 21-22 backward (library GEMMs + ATen glue: 13 and ~35).  Device-busy time of a
generator's forward + backward fell from 2.26 to 1.35 ms, the train step did not move: in the step every dependent tiny
launch costs >= 5 us of stream time whatever it computes.  What this buys is ownership: nothing here
touches a BLAS library, the generated kernels do not depend on which one PyTorch picked, and the product no longer
changes PyTorch's process-wide BLAS preference."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _p(t, off=0):
    return C.c_void_p(t.data_ptr() + 4 * off)


def gemm(A, a_off, sam, sak, B, b_off, sbn, sbk, Cm, c_off, scm, M, N, K, *, bias=None, aux=None, aux_out=None, batch=1,
         batA=0, batB=0, batC=0, alpha=1.0, epi=0, ksplit=1, part=0):
    """C[b](m, n) = epi(alpha * sum_k A[b](m, k) B[b](n, k)) on fp32 tensors; offsets and strides in elements."""
    L.check(L.load().eovae_gen_gemm(_p(A, a_off), _p(B, b_off), _p(Cm, c_off), L.ptr(bias), L.ptr(aux), L.ptr(aux_out), batch,
                                    M, N, K, sam, sak, sbn, sbk, scm, batA, batB, batC, alpha, epi, ksplit, part, L.stream()),
            'eovae_gen_gemm')


def _colsum(x, rows, n):
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    L.check(L.load().eovae_gen_colsum(L.ptr(x), L.ptr(out), rows, n, n, L.stream()), 'eovae_gen_colsum')
    return out


def _ksplit(M, N, K):
    """K slices of >= 128 (one fetch round of the kernel) so that a [M, N] output of 32 x 64 tiles fills the chip."""
    tiles = ((M + 31) // 32) * ((N + 63) // 64)
    return max(1, min(K // 128, 320 // max(tiles, 1)))


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for a [rows, in] (or [in]) fp32 input: one launch forward; dX, dW, db = three launches backward."""

    @staticmethod
    def forward(ctx, x, w, b):
        L.require_gpu(x, 'generator linear')
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        w = w.contiguous()
        rows, k = x2.shape
        n = w.shape[0]
        y = torch.empty(rows, n, dtype=torch.float32, device=x.device)
        gemm(x2, 0, k, 1, w, 0, k, 1, y, 0, n, rows, n, k, bias=b)
        ctx.save_for_backward(x2, w)
        ctx.shape, ctx.has_bias = shape, b is not None
        return y.view(*shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        rows, k = x2.shape
        n = w.shape[0]
        dy = dy.reshape(rows, n).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(rows, k, dtype=torch.float32, device=dy.device)
            gemm(dy, 0, n, 1, w, 0, 1, k, dx, 0, k, rows, k, n)                  # dx(r, j) = sum_o dy(r, o) W(o, j)
            dx = dx.view(ctx.shape)
        if ctx.needs_input_grad[1]:
            dw = torch.empty(n, k, dtype=torch.float32, device=dy.device)
            gemm(dy, 0, 1, n, x2, 0, 1, k, dw, 0, k, n, k, rows)                 # dW(o, j) = sum_r dy(r, o) x(r, j)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, rows, n)
        return dx, dw, db


def kernels_apply(x, *params):
    """Do the HIP kernels of this file take these tensors?  True: x is an fp32 tensor on the GPU (and then every parameter
    must be fp32 on the same device -- the kernels read raw pointers as fp32; anything else raises instead of being
    misread).  False: x is not -- a CPU tensor (the generator-only Stage-1 distillation needs no GPU) or another dtype
    -- and the caller uses the PyTorch-op formulation of the same layer (dynamic_conv._encoder_layer / nn.Linear)."""
    if not (x.is_cuda and x.dtype == torch.float32):
        return False
    for t in params:
        if t is not None and (t.dtype != torch.float32 or t.device != x.device):
            raise L.EovaeError(f'weight generator: a parameter is {t.dtype} on {t.device}, the input fp32 on {x.device}; the '
                               'generator kernels read fp32 on one device (keep the hyper-network in fp32: model.float())')
    return True


def linear(x, w, b=None):
    if not kernels_apply(x, w, b):
        return torch.nn.functional.linear(x, w, b)
    return LinearFn.apply(x, w, b)


class EncoderLayerFn(torch.autograd.Function):
    """nn.TransformerEncoderLayer.forward for an unbatched [L, d] input: post-norm, exact GELU, dropout 0 (the
    configuration of dynamic_conv.py:86-101), on the layer's own parameters.

    keep_from > 0 (last layer of a generator): only output rows keep_from .. L - 1 are computed -- the generators read
    nothing but the wavelength rows and the bias row (dynamic_conv.py:121-130, 173-183) and every op behind the
    attention scores is row-wise."""

    @staticmethod
    def forward(ctx, x, keep_from, heads, eps1, eps2, w_in, b_in, w_o, b_o, g1, be1, w1, bf1, w2, bf2, g2, be2):
        L.require_gpu(x, 'generator layer')
        lib = L.load()
        st = L.stream()
        x = x.contiguous()
        seq, d = x.shape
        dh = d // heads
        nq = seq - keep_from
        ff = w1.shape[0]
        dev = x.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        w_in, w_o, w1, w2 = w_in.contiguous(), w_o.contiguous(), w1.contiguous(), w2.contiguous()
        # 1. qkv = x W_in^T + b_in                                  [L, 3d]
        qkv = new(seq, 3 * d)
        gemm(x, 0, d, 1, w_in, 0, d, 1, qkv, 0, 3 * d, seq, 3 * d, d, bias=b_in)
        # 2. S_h = Q_h K_h^T / sqrt(dh) for the kept query rows      [h, nq, L]
        p = new(heads, nq, seq)
        gemm(qkv, keep_from * 3 * d, 3 * d, 1, qkv, d, 3 * d, 1, p, 0, seq, nq, seq, dh, batch=heads, batA=dh, batB=dh,
             batC=nq * seq, alpha=float(dh) ** -0.5)
        # 3. P = softmax rows (in place)
        L.check(lib.eovae_softmax_rows(L.ptr(p), heads * nq, seq, st), 'eovae_softmax_rows')
        # 4. O(:, h) = P_h V_h, heads side by side                   [nq, d]
        o = new(nq, d)
        gemm(p, 0, seq, 1, qkv, 2 * d, 1, 3 * d, o, 0, d, nq, dh, seq, batch=heads, batA=nq * seq, batB=dh, batC=dh)
        # 5. out_proj as split-K partial sums; 6. x1 = LN1(x + O W_o^T + b_o) folds them
        ks = _ksplit(nq, d, d)
        parts = new(ks, nq, d)
        gemm(o, 0, d, 1, w_o, 0, d, 1, parts, 0, d, nq, d, d, ksplit=ks, part=nq * d)
        x1, xh1, rs1 = new(nq, d), new(nq, d), new(nq)
        L.check(lib.eovae_gen_ln_fwd(L.ptr(parts), ks, nq * d, L.ptr(b_o), _p(x, keep_from * d), L.ptr(g1), L.ptr(be1),
                                     L.ptr(x1), L.ptr(xh1), L.ptr(rs1), nq, d, eps1, st), 'eovae_gen_ln_fwd')
        # 7. FFN: a = GELU(pre), pre = x1 W_1^T + b_1                [nq, ff]
        pre, act = new(nq, ff), new(nq, ff)
        gemm(x1, 0, d, 1, w1, 0, d, 1, act, 0, ff, nq, ff, d, bias=bf1, aux_out=pre, epi=1)
        # 8. y = LN2(x1 + a W_2^T + b_2), W_2 product split over K
        ks2 = _ksplit(nq, d, ff)
        parts2 = new(ks2, nq, d)
        gemm(act, 0, ff, 1, w2, 0, ff, 1, parts2, 0, d, nq, d, ff, ksplit=ks2, part=nq * d)
        y, xh2, rs2 = new(nq, d), new(nq, d), new(nq)
        L.check(lib.eovae_gen_ln_fwd(L.ptr(parts2), ks2, nq * d, L.ptr(bf2), L.ptr(x1), L.ptr(g2), L.ptr(be2), L.ptr(y),
                                     L.ptr(xh2), L.ptr(rs2), nq, d, eps2, st), 'eovae_gen_ln_fwd')
        ctx.save_for_backward(x, qkv, p, o, xh1, rs1, x1, pre, act, xh2, rs2, w_in, w_o, g1, w1, w2, g2)
        ctx.dims = (seq, d, heads, dh, nq, ff, keep_from)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, qkv, p, o, xh1, rs1, x1, pre, act, xh2, rs2, w_in, w_o, g1, w1, w2, g2 = ctx.saved_tensors
        seq, d, heads, dh, nq, ff, keep_from = ctx.dims
        lib = L.load()
        st = L.stream()
        dev = dy.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dy = dy.contiguous()
        alpha = float(dh) ** -0.5
        # LN2: d2 = grad of (x1 + f);  d_g2, d_be2
        d2, dg2, dbe2 = new(nq, d), new(d), new(d)
        L.check(lib.eovae_gen_ln_bwd(L.ptr(dy), None, 0, 0, None, L.ptr(xh2), L.ptr(rs2), L.ptr(g2), L.ptr(d2), L.ptr(dg2),
                                     L.ptr(dbe2), nq, d, st), 'eovae_gen_ln_bwd')
        # FFN second Linear: d_b2, d_W2(o, j) = sum_r d2(r, o) a(r, j)
        dbf2 = _colsum(d2, nq, d)
        dw2 = new(d, ff)
        gemm(d2, 0, 1, d, act, 0, 1, ff, dw2, 0, ff, d, ff, nq)
        # d_pre = (d2 W_2) o GELU'(pre)
        dpre = new(nq, ff)
        gemm(d2, 0, d, 1, w2, 0, 1, ff, dpre, 0, ff, nq, ff, d, aux=pre, epi=2)
        dbf1 = _colsum(dpre, nq, ff)
        dw1 = new(ff, d)
        gemm(dpre, 0, 1, ff, x1, 0, 1, d, dw1, 0, d, ff, d, nq)
        # d_x1 = d2 + d_pre W_1 (split over K; LN1's backward folds the partial sums)
        ks = _ksplit(nq, d, ff)
        parts = new(ks, nq, d)
        gemm(dpre, 0, ff, 1, w1, 0, 1, d, parts, 0, d, nq, d, ff, ksplit=ks, part=nq * d)
        d1, dg1, dbe1 = new(nq, d), new(d), new(d)
        L.check(lib.eovae_gen_ln_bwd(L.ptr(d2), L.ptr(parts), ks, nq * d, L.ptr(d2), L.ptr(xh1), L.ptr(rs1), L.ptr(g1),
                                     L.ptr(d1), L.ptr(dg1), L.ptr(dbe1), nq, d, st), 'eovae_gen_ln_bwd')
        # out_proj: d_b_o, d_W_o(o, j) = sum_r d1(r, o) O(r, j), d_O = d1 W_o
        dbo = _colsum(d1, nq, d)
        dwo = new(d, d)
        gemm(d1, 0, 1, d, o, 0, 1, d, dwo, 0, d, d, d, nq)
        do = new(nq, d)
        gemm(d1, 0, d, 1, w_o, 0, 1, d, do, 0, d, nq, d, d)
        # attention: d_P = d_O V^T, d_V = P^T d_O, d_S = softmax', d_Q = a d_S K, d_K = a d_S^T Q
        dqkv = torch.zeros(seq, 3 * d, dtype=torch.float32, device=dev) if keep_from else new(seq, 3 * d)
        dp = new(heads, nq, seq)
        gemm(do, 0, d, 1, qkv, 2 * d, 3 * d, 1, dp, 0, seq, nq, seq, dh, batch=heads, batA=dh, batB=dh, batC=nq * seq)
        gemm(p, 0, 1, seq, do, 0, 1, d, dqkv, 2 * d, 3 * d, seq, dh, nq, batch=heads, batA=nq * seq, batB=dh, batC=dh)
        L.check(lib.eovae_softmax_bwd_rows(L.ptr(p), L.ptr(dp), heads * nq, seq, st), 'eovae_softmax_bwd_rows')
        gemm(dp, 0, seq, 1, qkv, d, 1, 3 * d, dqkv, keep_from * 3 * d, 3 * d, nq, dh, seq, batch=heads, batA=nq * seq,
             batB=dh, batC=dh, alpha=alpha)
        gemm(dp, 0, 1, seq, qkv, keep_from * 3 * d, 1, 3 * d, dqkv, d, 3 * d, seq, dh, nq, batch=heads, batA=nq * seq,
             batB=dh, batC=dh, alpha=alpha)
        # in_proj: d_b_in, d_W_in, d_x (+ the residual branch d1 on the kept rows)
        dbin = _colsum(dqkv, seq, 3 * d)
        dwin = new(3 * d, d)
        gemm(dqkv, 0, 1, 3 * d, x, 0, 1, d, dwin, 0, d, 3 * d, d, seq)
        if keep_from == 0:
            dx = d1
            gemm(dqkv, 0, 3 * d, 1, w_in, 0, 1, d, dx, 0, d, seq, d, 3 * d, epi=3)
        else:
            dx = new(seq, d)
            gemm(dqkv, 0, 3 * d, 1, w_in, 0, 1, d, dx, 0, d, seq, d, 3 * d)
            dx[keep_from:] += d1
        return (dx, None, None, None, None, dwin, dbin, dwo, dbo, dg1, dbe1, dw1, dbf1, dw2, dbf2, dg2, dbe2)


def encoder_layer(layer, x, keep_from=0):
    """`layer`: an nn.TransformerEncoderLayer in the plain configuration (dynamic_conv._layer_is_plain)."""
    att = layer.self_attn
    if not kernels_apply(x, att.in_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias, layer.norm1.weight,
                         layer.norm1.bias, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight,
                         layer.linear2.bias, layer.norm2.weight, layer.norm2.bias):
        from .dynamic_conv import _encoder_layer      # the PyTorch-op formulation of the same layer
        return _encoder_layer(layer, x, keep_from)
    return EncoderLayerFn.apply(x, int(keep_from), att.num_heads, float(layer.norm1.eps), float(layer.norm2.eps),
                                att.in_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias,
                                layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
                                layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias)
