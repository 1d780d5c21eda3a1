"""torch.autograd bindings of the HIP kernels (C-ABI in include/eovae_hip.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape.  Every
forward/backward below is one or more calls into libeovae_hip.so on the current stream."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L

MODE_NORMAL, MODE_DOWN, MODE_UP, MODE_ZINS = 0, 1, 2, 3
_MODES = {'normal': MODE_NORMAL, 'down': MODE_DOWN, 'up': MODE_UP}

_ws_cache: dict = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per device (kernels on one stream run in order)."""
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------


def _packed_dims(cout, cin, ks, mode, flip):
    kp, mp = C.c_int(), C.c_int()
    L.check(L.load().eovae_conv_packed_dims(cout, cin, ks, mode, flip, C.byref(kp), C.byref(mp)),
            'eovae_conv_packed_dims')
    return kp.value, mp.value


def pack_weights(w: torch.Tensor, mode: int, flip: bool, scale: float = 1.0):
    cout, cin, ks, _ = w.shape
    kp, mp = _packed_dims(cout, cin, ks, mode, int(flip))
    wp = torch.empty(ks * ks * kp * mp, dtype=torch.float32, device=w.device)
    L.check(L.load().eovae_conv_pack_weights(L.ptr(w), L.ptr(wp), cout, cin, ks, mode, int(flip),
                                             scale, L.stream()), 'eovae_conv_pack_weights')
    return wp, kp, mp


def _out_hw(h, w, mode):
    if mode == MODE_NORMAL:
        return h, w
    if mode == MODE_DOWN:
        return (h - 2) // 2 + 1, (w - 2) // 2 + 1
    return 2 * h, 2 * w


def conv_raw(x, wp, kp, mp, bias, residual, cout, ks, mode, out_scale=1.0):
    n, cin, h, w = x.shape
    ho, wo = _out_hw(h, w, mode)
    y = torch.empty(n, cout, ho, wo, dtype=torch.float32, device=x.device)
    L.check(L.load().eovae_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(residual), L.ptr(y),
                                      n, cin, h, w, cout, ho, wo, ks, mode, kp, mp, out_scale,
                                      L.stream()), 'eovae_conv2d_fwd')
    return y


class Conv2dFn(torch.autograd.Function):
    """y = conv(x, w) + b (+ residual); mode in {'normal','down','up'} (see csrc/conv_f32.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, mode):
        L.require_gpu(x, 'conv2d')
        x, weight = _c(x), _c(weight)
        m = _MODES[mode]
        cout, cin, ks, _ = weight.shape
        if x.shape[1] != cin:
            raise L.EovaeError(f'conv2d: input has {x.shape[1]} channels, weight expects {cin}')
        wp, kp, mp = pack_weights(weight, m, False)
        res = _c(residual) if residual is not None else None
        y = conv_raw(x, wp, kp, mp, _c(bias) if bias is not None else None, res, cout, ks, m)
        ctx.save_for_backward(x, weight)
        ctx.mode, ctx.has_bias, ctx.has_res = m, bias is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _c(dy)
        m = ctx.mode
        cout, cin, ks, _ = weight.shape
        n, _, h, w = x.shape
        lib = L.load()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wp, kp, mp = pack_weights(weight, m, True)
            if m == MODE_NORMAL:
                dx = conv_raw(dy, wp, kp, mp, None, None, cin, ks, MODE_NORMAL)
            elif m == MODE_DOWN:
                dx = conv_raw(dy, wp, kp, mp, None, None, cin, ks, MODE_ZINS)
                if dx.shape[2] != h or dx.shape[3] != w:  # odd input sizes: crop / pad never needed for even
                    raise L.EovaeError('conv2d(down): odd spatial sizes are not supported')
            else:  # up: gradient w.r.t. the virtual x2 image, then sum each 2x2 block
                du = conv_raw(dy, wp, kp, mp, None, None, cin, ks, MODE_NORMAL)
                dx = torch.empty_like(x)
                L.check(lib.eovae_upsample2x_bwd(L.ptr(du), L.ptr(dx), n, cin, h, w, L.stream()),
                        'eovae_upsample2x_bwd')
        if ctx.needs_input_grad[1]:
            dwp = torch.zeros(ks * ks * cout * cin, dtype=torch.float32, device=x.device)
            L.check(lib.eovae_conv2d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), n, cin, h, w, cout,
                                           dy.shape[2], dy.shape[3], ks, m, L.stream()),
                    'eovae_conv2d_wgrad')
            dw = torch.empty_like(weight)
            L.check(lib.eovae_conv_unpack_wgrad(L.ptr(dwp), L.ptr(dw), cout, cin, ks, L.stream()),
                    'eovae_conv_unpack_wgrad')
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(cout, dtype=torch.float32, device=x.device)
            L.check(lib.eovae_bias_grad(L.ptr(dy), L.ptr(db), n, cout, dy.shape[2] * dy.shape[3],
                                        L.stream()), 'eovae_bias_grad')
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None


def conv2d(x, weight, bias=None, residual=None, mode='normal'):
    return Conv2dFn.apply(x, weight, bias, residual, mode)


# ------------------------------------------------------------------------------------------
# GroupNorm (+ swish)
# ------------------------------------------------------------------------------------------


class GroupNormSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu):
        L.require_gpu(x, 'group_norm')
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        n, c, h, w = x.shape
        lib = L.load()
        ws = workspace(lib.eovae_gn_workspace_bytes(n, c, h * w, groups), x.device)
        y = torch.empty_like(x)
        stats = torch.empty(n * groups * 2, dtype=torch.float32, device=x.device)
        L.check(lib.eovae_gn_silu_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(stats),
                                      n, c, h * w, groups, eps, int(silu), L.ptr(ws), L.stream()),
                'eovae_gn_silu_fwd')
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.groups, ctx.silu = groups, silu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        dy = _c(dy)
        n, c, h, w = x.shape
        lib = L.load()
        ws = workspace(lib.eovae_gn_workspace_bytes(n, c, h * w, ctx.groups), x.device)
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                      None, L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), n, c, h * w,
                                      ctx.groups, int(ctx.silu), L.ptr(ws), L.stream()),
                'eovae_gn_silu_bwd')
        return dx, dgamma, dbeta, None, None, None


def group_norm_silu(x, gamma, beta, groups=32, eps=1e-6, silu=True):
    return GroupNormSiluFn.apply(x, gamma, beta, groups, eps, silu)
