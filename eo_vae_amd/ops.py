"""torch.autograd bindings of the HIP kernels (C-ABI in include/eovae_hip.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape.  Every
forward/backward below is one or more calls into libeovae_hip.so on the current stream."""
from __future__ import annotations

import ctypes as C
import os

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


_ZERO_POOL = {}
_ZERO_POOL_ELEMS = 1 << 18


def zeros_small(numel: int, dtype, device) -> torch.Tensor:
    """A zeroed 1-d tensor for a small accumulator (GroupNorm sums, bias gradients): a never-reused slice of a pool
    that ONE fill kernel zeroed, instead of a 5-9 us fill launch per accumulator (about 110 per step).  A slice is
    handed out once; a used-up pool is dropped (it lives on while slices of it do).  NOTE for bias gradients: autograd
    keeps the slice as the parameter's .grad, i.e. a view into a 1 MiB pool that stays alive with it -- FlatParams
    gathers and drops those gradients every step; a caller that keeps or pickles .grad tensors should clone them."""
    if numel > _ZERO_POOL_ELEMS // 8:
        return torch.zeros(numel, dtype=dtype, device=device)
    # one pool per (device, dtype, STREAM, inference mode): a pool is zero-filled on the stream current at its creation,
    # and a pool created under torch.inference_mode() is an inference tensor that a later train-mode step must not reuse
    key = (device.type, device.index, dtype, torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0,
           torch.is_inference_mode_enabled())
    pool, off = _ZERO_POOL.get(key, (None, 0))
    step = (numel + 63) // 64 * 64            # keep every slice 256-byte aligned
    if pool is None or off + step > pool.numel():
        pool, off = torch.zeros(_ZERO_POOL_ELEMS, dtype=dtype, device=device), 0
    _ZERO_POOL[key] = (pool, off + step)
    return pool[off:off + numel]


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


class KernelTimer:
    """Opt-in HIP-event bracket around every launch of one kernel family on the launch stream
    (bench.py: roofline.achieved = algorithmic flops / measured kernel time)."""

    def __init__(self):
        self.records = []  # (start_event, end_event, algorithmic_flops)

    def bracket(self, flops, nbytes=0.0, kind=0):
        """kind: which kernel of the family takes the launch (3x3 f16c8: 1 = conv3x3_f16c8_w4_kernel, 0 = the rest)."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((a, b, flops, nbytes, kind))
        return a, b

    def bytes_total(self, kind=None):
        return sum(r[3] for r in self.records if kind is None or r[4] == kind)

    def totals(self, kind=None):
        torch.cuda.synchronize()
        recs = [r for r in self.records if kind is None or r[4] == kind]
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        return ms, sum(r[2] for r in recs), len(recs)


CONV3X3_TIMER: KernelTimer | None = None  # set by bench.py during the timed region
WGRAD3X3_TIMER: KernelTimer | None = None  # same for the 3x3 weight-gradient launches (second roofline object)


def conv_raw(x, wp, kp, mp, bias, residual, cout, ks, mode, out_scale=1.0):
    n, cin, h, w = x.shape
    ho, wo = _out_hw(h, w, mode)
    y = torch.empty(n, cout, ho, wo, dtype=torch.float32, device=x.device)
    timed = CONV3X3_TIMER is not None and ks == 3 and mode != MODE_DOWN
    if timed:
        # algorithmic flops of the layer this launch implements (MODE_ZINS = adjoint of the
        # stride-2 layer: its algorithmic work is the stride-2 layer's, a quarter of the tile work)
        pix = (ho * wo) // 4 if mode == MODE_ZINS else ho * wo
        ev0, ev1 = CONV3X3_TIMER.bracket(2.0 * n * pix * cout * cin * 9, 4.0 * (x.numel() + y.numel()))
        ev0.record()
    L.check(L.load().eovae_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(residual), L.ptr(y),
                                      n, cin, h, w, cout, ho, wo, ks, mode, kp, mp, out_scale,
                                      L.stream()), 'eovae_conv2d_fwd')
    if timed:
        ev1.record()
    return y


# Arithmetic of the 3x3 stride-1 convolutions (99 % of the step's flops):
#   'f32'    exact fp32 on v_mfma_f32_32x32x2_f32 (csrc/conv_f32.hip)
#   'bf16x3' split-bf16, 3 bf16 MFMAs per step, fp32-grade (~2^-17)  (csrc/conv_bf16x3.hip)
#   'bf16'   plain bf16 operands, fp32 accumulate (fast mode; does NOT meet the 1e-3 parity bar)
#   'f16c8'  f16 product + one block-scaled fp8 product carrying both first-order corrections: 2 MFMA
#            units per algorithmic product instead of 3, whole-model error 1.2e-4 (csrc/conv_f16c8.hip);
#            3x3 forward / data gradient with K % 32 == 0, everything else runs as 'bf16x3'
DEFAULT_PRECISION = os.environ.get('EOVAE_CONV_PRECISION', 'f16c8')   # the parity path (fastest inside the bar)
CONV_PRECISION = DEFAULT_PRECISION
_PREC = {'bf16x3': 3, 'bf16': 1, 'f16c8': 3}


class precision_scope:
    """`with ops.precision_scope('bf16'):` -- arithmetic mode for the convolutions / GEMMs issued inside
    the block; the previous mode comes back on exit (also on an exception).  Conv2dFn records the mode
    of its forward and replays its backward in the same mode, so a scope that ends between forward
    and backward cannot mix code paths."""

    def __init__(self, mode):
        if mode not in ('f32', 'bf16x3', 'bf16', 'f16c8'):
            raise ValueError(f'unknown conv precision {mode!r}')
        self.mode = mode

    def __enter__(self):
        global CONV_PRECISION
        self.prev, CONV_PRECISION = CONV_PRECISION, self.mode
        return self

    def __exit__(self, *exc):
        global CONV_PRECISION
        CONV_PRECISION = self.prev
        return False


def _use_f16c8(kreal):
    return CONV_PRECISION == 'f16c8' and kreal % 32 == 0


def absmax_bits(t):
    """[1] int32 device tensor holding the float bits of max|t| (no host sync): the power-of-two
    pre-scale of an f16c8 data-gradient call is derived from it inside the kernel."""
    out = torch.empty(1, dtype=torch.int32, device=t.device)
    L.check(L.load().eovae_absmax_bits(L.ptr(t), t.numel(), L.ptr(out), L.stream()), 'eovae_absmax_bits')
    return out


# A kernel that writes a gradient tensor can report max|t| on the way (GroupNorm backward does): the value
# rides on the tensor object as (bits tensor, _version).  PyTorch keeps a tensor's Python object alive for as
# long as the C++ tensor lives, so the tag survives the hand-over between autograd nodes; an in-place
# accumulation by the engine bumps _version and silently drops it (the consumer then measures again).
def _new_amax(t):
    return torch.empty(1, dtype=torch.int32, device=t.device) if CONV_PRECISION == 'f16c8' else None


def _tag_amax(t, amax):
    if amax is not None:
        t._eovae_amax = (amax, t._version)


# GroupNorm + SiLU can hand its result to the next 3x3 convolution as f16c8 OPERAND RECORDS instead of fp32
# (csrc/groupnorm.hip gn_apply_packed_kernel): a float32 tensor of the usual [N, C, H, W] shape whose BYTES are the
# records (N*C*H*W dwords either way).  It is only meaningful to Conv2dFn (plain 3x3, f16c8).  The hand-over is a
# tensor SUBCLASS, so the marking survives detach() / clone() / views / checkpoint recompute (a Python attribute did
# not, and the bytes were then read as fp32 without any error), and every torch operation that would compute on the
# bytes raises instead of returning garbage.
class PackedRecords(torch.Tensor):
    """f16c8 operand records in a float32 shell: consumable by ops.conv2d (plain 3x3, f16c8 mode) only."""

    _PASS = frozenset({'detach', 'clone', 'contiguous', 'is_contiguous', 'data_ptr', 'size', 'stride', 'dim', 'numel',
                       'element_size', 'storage_offset', 'untyped_storage', 'requires_grad_', 'retain_grad', 'register_hook',
                       'view_as', 'is_pinned', 'get_device', 'record_stream', '__get__', '__repr__', '__str__', '__format__',
                       '__hash__', '__len__', '__reduce_ex__', '__deepcopy__', '__dir__', 'is_floating_point', '_backward_hooks',
                       'backward', 'grad_fn'})

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, '__name__', str(func))
        if name in ('__repr__', '__str__', '__format__'):
            t = args[0]
            return f'PackedRecords(shape={tuple(t.shape)}, device={t.device}: f16c8 operand records, not fp32 values)'
        if name not in cls._PASS:
            raise L.EovaeError(f'{name}() on f16c8 operand records (GroupNorm packed output): the bytes are not fp32 values; '
                               'only the plain 3x3 f16c8 convolution (ops.conv2d) consumes them -- call '
                               'group_norm_silu(..., packed=False) for an fp32 result')
        return super().__torch_function__(func, types, args, kwargs or {})


def mark_packed(t):
    return t if isinstance(t, PackedRecords) else t.as_subclass(PackedRecords)


def is_packed(t):
    return isinstance(t, PackedRecords)


def packed_ok(channels):
    """Can a GroupNorm output with this many channels be handed over as operand records?"""
    return CONV_PRECISION == 'f16c8' and channels % 64 == 0 and PACKED_ACTIVATIONS


PACKED_ACTIVATIONS = os.environ.get('EOVAE_PACKED_ACT', '1') != '0'


# Round 5: MX operand records ("format 2", csrc/groupnorm.hip gn_apply_mx_kernel) -- the same 4 bytes per element in two
# planes: 8 x f16 per pixel and channel octet, and per pixel and 32 channels the block-scaled FP6 (e2m3) operands of the
# forward convolution's correction product.  Their consumers are the forward of a plain 3x3 layer (MX-FP6 correction
# product, ~1.6 x the fp8 rate) and its weight gradient in the f16 form of the row-streaming kernel; the GroupNorm writes
# them only when told which layer consumes its result (`cout=`) and the library says that layer takes them
# (eovae_records_mx_ok).  A subclass again: the format survives detach() / clone() / views like the marking itself.
class PackedRecordsMX(PackedRecords):
    """MX operand records (f16 plane + MX-FP6 plane) in a float32 shell."""


def mark_packed_mx(t):
    return t if isinstance(t, PackedRecordsMX) else t.as_subclass(PackedRecordsMX)


def packed_format(t):
    """0 = fp32 values, 1 = f16c8 operand records, 2 = MX operand records (the `xpacked` code of the C entries)."""
    return 2 if isinstance(t, PackedRecordsMX) else (1 if isinstance(t, PackedRecords) else 0)


def records_format_for(x, cout):
    """Format a GroupNorm over x writes for a consumer that is a plain 3x3 layer with `cout` output channels (None / 0:
    unknown consumer -> format 1, which every record consumer reads)."""
    if not cout or CONV_PRECISION != 'f16c8':
        return 1
    n, c, h, w = x.shape
    return 2 if L.load().eovae_records_mx_ok(n, c, h, w, int(cout)) == 1 else 1


def mx_planes(t):
    """(f16 plane, MX-FP6 plane) of MX operand records, as int32 views [N, C/8, HW, 4] and [N, C/32, 4, HW, 4] (tests, tools)."""
    if not isinstance(t, PackedRecordsMX):
        raise L.EovaeError('mx_planes: not MX operand records')
    n, c, h, w = t.shape
    with torch._C.DisableTorchFunctionSubclass():
        raw = t.detach().as_subclass(torch.Tensor).reshape(-1).view(torch.int32)
    half = raw.numel() // 2
    return raw[:half].view(n, c // 8, h * w, 4), raw[half:].view(n, c // 32, 4, h * w, 4)


def records_f16(t):
    """The f16 leading parts an operand-record tensor (either format) holds, as a float32 [N, C, H, W] tensor (tests, tools)."""
    fmt = packed_format(t)
    if not fmt:
        raise L.EovaeError('records_f16: not operand records')
    n, c, h, w = t.shape
    with torch._C.DisableTorchFunctionSubclass():
        raw = t.detach().as_subclass(torch.Tensor).reshape(-1).view(torch.int32)
    if fmt == 2:
        f = raw[:raw.numel() // 2].view(n, c // 8, h * w, 4)
    else:
        f = raw.view(n, c // 8, h * w, 8)[..., :4].contiguous()
    return f.view(torch.float16).view(n, c // 8, h * w, 8).permute(0, 1, 3, 2).reshape(n, c, h, w).float()


# Every 3x3 convolution of the model is followed by a GroupNorm(32) over its output (layers.py:91-92, 106-108;
# model.py norm_out): the f16c8 forward kernel adds that GroupNorm's (sum, sum of squares) per (image, group) from
# its epilogue, and the tag below lets group_norm_silu() skip its statistics pass over the tensor.
GN_FUSED_STATS = os.environ.get('EOVAE_GN_FUSED', '1') != '0'
GN_GROUPS_HINT = 32
_LAST_GN_SUMS = None   # handed from Conv2dFn.forward to conv2d(), which tags the tensor autograd returns


def _gn_sums_buffer(n, cout, device):
    """Zeroed [n, 32, 2] float64 accumulator when the epilogue can produce the statistics, else None."""
    cpg = cout // GN_GROUPS_HINT if cout % GN_GROUPS_HINT == 0 else 0
    if not GN_FUSED_STATS or cpg < 4 or cpg > 128 or cpg & (cpg - 1):
        return None
    return zeros_small(n * GN_GROUPS_HINT * 2, torch.float64, device).view(n, GN_GROUPS_HINT, 2)


def gn_sums_of(x, groups):
    """The producer's (sum, sumsq) accumulator for GroupNorm(groups) over x, if x still is what was summed."""
    tag = getattr(x, '_eovae_gn_sums', None)
    if tag is not None and tag[1] == groups and tag[2] == x._version and tag[0].shape[0] == x.shape[0] \
            and tag[0].device == x.device:
        return tag[0]
    return None


# f16c8 operands are f16 / e5m2 values: GroupNorm + SiLU outputs (operand records) are bounded by construction,
# gradients are pre-scaled from their amax, and -- RANGE_GUARD, on by default -- so is every OTHER fp32 input of a 3x3
# layer (the residual stream in front of a Downsample / Upsample layer, the latent in front of the decoder's conv_in):
# one absmax pass per such tensor (7 per step, ~0.3 ms) buys the removal of the |x| < 65 504 assumption.
RANGE_GUARD = os.environ.get('EOVAE_F16C8_RANGE_GUARD', '1') != '0'
# fp32 weight-gradient kernel for the first / last layer with <= 4 channels on the thin side (csrc/conv_thin.hip); 0 = the
# matrix-core kernels as for every other layer (A/B)
THIN_WGRAD = os.environ.get('EOVAE_THIN_WGRAD', '1') != '0'
# plain f16c8 weight gradients through per-block partial tiles + a fold (deterministic, no atomics); 0 = atomics into a
# zero-filled accumulator + unpack (A/B)
WGRAD_DIRECT = os.environ.get('EOVAE_WGRAD_DIRECT', '1') != '0'


# The guard's scale only has to keep |x| * scale inside f16 / e5m2: where x still carries the GroupNorm accumulators of
# the convolution that wrote it (every f16c8 forward output does), sqrt(max sum of squares) >= max|x| serves, from a
# 512-value reduction instead of a pass over x (six per step, 0.36 ms).  0: exact maxima by reduction passes (A/B).
AMAX_FROM_SUMS = os.environ.get('EOVAE_AMAX_FROM_SUMS', '1') != '0'


def range_amax(x):
    """absmax bits (or an upper bound's) for the pre-scale of an fp32, not GroupNorm-normalised 3x3-layer input
    (None: guard off)."""
    if not (RANGE_GUARD and CONV_PRECISION == 'f16c8'):
        return None
    tag = getattr(x, '_eovae_amax', None)
    if tag is not None and tag[1] == x._version and tag[0].device == x.device:
        return tag[0]
    st = getattr(x, '_eovae_gn_sums', None)
    if AMAX_FROM_SUMS and st is not None and st[2] == x._version and st[0].device == x.device and st[0].shape[0] == x.shape[0]:
        bits = torch.empty(1, dtype=torch.int32, device=x.device)
        L.check(L.load().eovae_amax_bound_from_sums(L.ptr(st[0]), st[0].shape[0] * st[0].shape[1], L.ptr(bits), L.stream()),
                'eovae_amax_bound_from_sums')
        _tag_amax(x, bits)
        return bits
    return amax_of(x)


def amax_of(t):
    """absmax bits of t: the producer's tag when it is still valid, else one reduction pass."""
    tag = getattr(t, '_eovae_amax', None)
    if tag is not None and tag[1] == t._version and tag[0].device == t.device:
        return tag[0]
    bits = absmax_bits(t)
    _tag_amax(t, bits)     # the data gradient and the weight gradient of one layer ask for the same tensor
    return bits


def _pack3x3(w4, rows, cols, flip, f16c8):
    """Pack a [rows][cols][3][3] fp32 layer for the 3x3 kernels (split-bf16 tiles or f16c8 tiles)."""
    lib = L.load()
    if f16c8:
        wp = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(rows, cols, int(flip)), dtype=torch.uint8,
                         device=w4.device)
        L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(w4), L.ptr(wp), rows, cols, int(flip), 1.0, L.stream()),
                'eovae_conv3x3_f16c8_pack')
    else:
        wp = torch.empty(lib.eovae_conv3x3_bf16_packed_elems(rows, cols, int(flip)), dtype=torch.bfloat16,
                         device=w4.device)
        L.check(lib.eovae_conv3x3_bf16_pack(L.ptr(w4), L.ptr(wp), rows, cols, int(flip), 1.0, L.stream()),
                'eovae_conv3x3_bf16_pack')
    return wp


def _launch3x3(x, wp, bias, residual, y, n, cin_g, h, w, cout_gemm, ho, wo, mode, subpix, f16c8, amax=None,
               packed=0, gn_sums=None, out_amax=None):
    """out_amax (f16c8 only): a zeroed [1] int32 that receives the float bits of max|y| from the kernel's epilogue.
    packed: 0 fp32 input, 1 operand records, 2 MX operand records (packed_format)."""
    lib = L.load()
    if packed and not f16c8:
        raise L.EovaeError('conv2d: operand-record input reached a non-f16c8 kernel')
    if f16c8:
        L.check(lib.eovae_conv3x3_f16c8_fwd_mx(L.ptr(x), None, L.ptr(wp), L.ptr(bias), L.ptr(residual),
                                               L.ptr(y), L.ptr(amax), n, cin_g, h, w, cout_gemm, ho, wo, mode, subpix,
                                               int(packed), L.ptr(gn_sums), GN_GROUPS_HINT if gn_sums is not None else 0,
                                               L.ptr(out_amax), L.stream()), 'eovae_conv3x3_f16c8_fwd')
    else:
        L.check(lib.eovae_conv3x3_bf16_fwd(L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(residual), L.ptr(y), n,
                                           cin_g, h, w, cout_gemm, ho, wo, mode, _PREC[CONV_PRECISION], subpix,
                                           L.stream()), 'eovae_conv3x3_bf16_fwd')


def _launch_wgrad3x3(x, dy, dwp, db, n, cin, h, w, cout, ho, wo, mode, subpix, packed=0, xamax=None, dw=None):
    """3x3 weight gradient into dwp (pre-zeroed): f16c8 kernel in f16c8 mode, else split-bf16.
    xamax: absmax bits of an fp32 x that is not a GroupNorm output (range_amax), or None."""
    lib = L.load()
    amax = amax_of(dy) if CONV_PRECISION == 'f16c8' else None
    if CONV_PRECISION == 'f16c8' and not packed and xamax is None:
        xamax = range_amax(x)
    timed = WGRAD3X3_TIMER is not None
    if timed:
        # algorithmic flops of the layer as the reference executes it: 9 taps on the layer's output grid
        # (subpix 2 = the stride-2 layer: its output grid is the low-res one)
        ev0, ev1 = WGRAD3X3_TIMER.bracket(2.0 * n * ho * wo * cout * cin * 9, 4.0 * (x.numel() + dy.numel()))
        ev0.record()
    if CONV_PRECISION == 'f16c8' and dw is not None:
        # plain layer: partial tiles + a fold in a fixed order straight into dw (no zero-filled accumulator, no float
        # atomics, no unpack launch)
        ws = workspace(lib.eovae_conv3x3_f16c8_wgrad_dw_workspace(cin, cout), x.device)
        L.check(lib.eovae_conv3x3_f16c8_wgrad_dw(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(amax), L.ptr(xamax), n,
                                                 cin, h, w, cout, ho, wo, int(packed), L.ptr(ws), L.stream()),
                'eovae_conv3x3_f16c8_wgrad_dw')
    elif CONV_PRECISION == 'f16c8':
        L.check(lib.eovae_conv3x3_f16c8_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), L.ptr(db), L.ptr(amax), L.ptr(xamax), n,
                                              cin, h, w, cout, ho, wo, mode, subpix, int(packed), L.stream()),
                'eovae_conv3x3_f16c8_wgrad')
    elif packed:
        raise L.EovaeError('conv2d backward: operand-record input reached a non-f16c8 kernel')
    else:
        L.check(lib.eovae_conv3x3_bf16_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), L.ptr(db), n, cin, h, w, cout, ho,
                                             wo, mode, _PREC[CONV_PRECISION], subpix, L.stream()),
                'eovae_conv3x3_bf16_wgrad')
    if timed:
        ev1.record()


class WeightPrepack:
    """Forward and data-gradient f16c8 packs of EVERY plain 3x3 layer of a model from two launches per step
    (eovae_conv3x3_f16c8_pack_multi) instead of three launches per layer.  The packs are only trusted between
    `run()` and `release()` of the same training step (EOFluxVAE.training_step): nothing is cached across optimiser
    steps, and a weight whose storage moved since the descriptors were built makes `stale()` true."""

    def __init__(self, weights):
        import numpy as np
        lib = L.load()
        self.weights = [w for w in weights if w.is_cuda and w.dim() == 4 and w.shape[2] == 3 and w.shape[3] == 3
                        and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and w.dtype == torch.float32 and w.is_contiguous()]
        self.packs, rows = {}, []
        for w in self.weights:
            cout, cin = w.shape[0], w.shape[1]
            wp0 = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device=w.device)
            wp1 = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 1), dtype=torch.uint8, device=w.device)
            self.packs[w.data_ptr()] = (wp0, wp1)
            rows.append((w.data_ptr(), wp0.data_ptr(), wp1.data_ptr(), cout, cin))
        desc = np.zeros(len(rows), dtype=np.dtype([('w', '<u8'), ('p0', '<u8'), ('p1', '<u8'), ('co', '<i4'), ('ci', '<i4')]))
        for i, r in enumerate(rows):
            desc[i] = r
        self.ptrs = [r[0] for r in rows]
        self.desc = (torch.from_numpy(desc.view(np.uint8).copy()).to(self.weights[0].device) if rows else None)

    def stale(self):
        return [w.data_ptr() for w in self.weights] != self.ptrs

    def run(self):
        """Enqueue the two pack launches -- on the current stream, or (EOVAE_PREPACK_STREAM=1) on a side stream, in which case
        the first convolution that asks for a pack makes its stream wait for the packs' event."""
        global _ACTIVE_PREPACK, _PREPACK_STREAM
        if self.desc is None:
            return
        main = torch.cuda.current_stream()
        if PREPACK_SIDE_STREAM:
            if _PREPACK_STREAM is None:
                _PREPACK_STREAM = torch.cuda.Stream()
                from .parallel import SIDE_STREAMS
                SIDE_STREAMS.append(_PREPACK_STREAM)
            side = _PREPACK_STREAM
            side.wait_stream(main)      # the optimiser step that wrote the weights; last step's convolutions that read the packs
            with torch.cuda.stream(side):
                L.check(L.load().eovae_conv3x3_f16c8_pack_multi(L.ptr(self.desc), len(self.ptrs), 1.0, L.stream()),
                        'eovae_conv3x3_f16c8_pack_multi')
                self.event = side.record_event()
        else:
            L.check(L.load().eovae_conv3x3_f16c8_pack_multi(L.ptr(self.desc), len(self.ptrs), 1.0, L.stream()),
                    'eovae_conv3x3_f16c8_pack_multi')
            self.event = None
        # a pack is only handed out for the very tensor it was made from: same storage, shape and version (a
        # slice sharing offset 0, or an in-place edit between run() and the convolution, must not get this pack)
        self.versions = {w.data_ptr(): (tuple(w.shape), w._version) for w in self.weights}
        self.waited = set()
        _ACTIVE_PREPACK = self

    def lookup(self, weight):
        hit = self.packs.get(weight.data_ptr())
        if hit is None or self.versions.get(weight.data_ptr()) != (tuple(weight.shape), weight._version):
            return None
        if self.event is not None:       # a stream's first consumer makes it wait for the packs (later ones follow in order)
            cs = torch.cuda.current_stream()
            if cs.cuda_stream not in self.waited:
                cs.wait_event(self.event)
                self.waited.add(cs.cuda_stream)
        return hit

    event = None

    @staticmethod
    def release():
        global _ACTIVE_PREPACK
        _ACTIVE_PREPACK = None


_ACTIVE_PREPACK = None
_PREPACK_STREAM = None
PREPACK = os.environ.get('EOVAE_PREPACK', '1') != '0'
# EOVAE_PREPACK keeps both orientations of every plain 3x3 weight packed for the whole run (~8 bytes per weight element:
# 0.7 GB for the full-size model) instead of transient per-layer buffers.
# EOVAE_PREPACK_STREAM=1: the two pack launches on a side stream, overlapping the encoder generator's head.  Measured
# (profiles/r04n_prepack_stream_ab.log, same box, alternating): 97.4-97.6 ms on the main stream, 97.6-98.1 on the side stream
# -- the packs' weight reads then miss the L2 the convolutions find them in.  Off by default.
PREPACK_SIDE_STREAM = os.environ.get('EOVAE_PREPACK_STREAM', '0') != '0'


def pack_pair_f16c8(weight):
    """(forward, data-gradient) f16c8 weight packs of one plain 3x3 layer from ONE amax pass."""
    if _ACTIVE_PREPACK is not None:
        hit = _ACTIVE_PREPACK.lookup(weight)
        if hit is not None:
            return hit
    lib = L.load()
    cout, cin = weight.shape[0], weight.shape[1]
    wp0 = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device=weight.device)
    wp1 = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 1), dtype=torch.uint8, device=weight.device)
    L.check(lib.eovae_conv3x3_f16c8_pack_pair(L.ptr(weight), L.ptr(wp0), L.ptr(wp1), cout, cin, 1.0, L.stream()),
            'eovae_conv3x3_f16c8_pack_pair')
    return wp0, wp1


def conv3x3_lowp(x, weight, bias, residual, cout_gemm, mode, flip, amax=None, packed=0, wp=None):
    """3x3 stride-1 conv / data-gradient on the 16-bit matrix cores (weight in PyTorch layout).
    amax: absmax_bits(x) for a data-gradient call (x = dy) in f16c8 mode.  wp: an already packed weight."""
    cout, cin = weight.shape[0], weight.shape[1]
    n, cin_g, h, w = x.shape
    f8 = _use_f16c8(cin_g) and mode == MODE_NORMAL
    if f8 and flip and amax is None:
        amax = amax_of(x)
    if f8 and not flip and not packed and amax is None:
        amax = range_amax(x)
    if wp is None:
        wp = _pack3x3(weight, cout, cin, flip, f8)
    ho, wo = _out_hw(h, w, mode)
    y = torch.empty(n, cout_gemm, ho, wo, dtype=torch.float32, device=x.device)
    timed = CONV3X3_TIMER is not None
    if timed:
        pix = (ho * wo) // 4 if mode == MODE_ZINS else ho * wo
        kind = L.load().eovae_conv3x3_f16c8_which(n, cin_g, h, w, cout_gemm, ho, wo, mode, 0, int(packed)) if f8 else 0
        ev0, ev1 = CONV3X3_TIMER.bracket(2.0 * n * pix * cout_gemm * cin_g * 9,
                                         4.0 * (x.numel() + y.numel() + weight.numel()), kind)
        ev0.record()
    if packed and not f8:
        raise L.EovaeError('conv2d: operand-record input needs the f16c8 kernel (K % 32 == 0, mode normal)')
    global _LAST_GN_SUMS, _LAST_OUT_AMAX
    sums = _gn_sums_buffer(n, cout_gemm, x.device) if f8 and not flip else None
    oa = None   # (forward outputs are not tagged: most come from the one-wave-per-SIMD kernel, whose epilogue has no
                # register for a running maximum, and are read un-normalised by 6 of ~60 layers only)
    _launch3x3(x, wp, bias, residual, y, n, cin_g, h, w, cout_gemm, ho, wo, mode, 0, f8, None if packed else amax,
               packed, sums, oa)
    _LAST_GN_SUMS = sums
    _LAST_OUT_AMAX = oa
    if timed:
        ev1.record()
    return y


# "nearest x2 + 3x3" as four 2x2 sub-pixel kernels on the low-res grid (4/9 of the MFMA work);
# needs Cout % 128 == 0 (the 256/512-channel upsample layers of the model), else the plain path.
SUBPIX_UPSAMPLE = os.environ.get('EOVAE_SUBPIX', '1') != '0'


UP_WGRAD_ROWS = True   # False: the tile kernel's sub-pixel mode (tests, A/B)
EPILOGUE_AMAX = os.environ.get('EOVAE_EPILOGUE_AMAX', '1') != '0'   # max|y| from the convolution epilogues (0: reduction passes)
_LAST_OUT_AMAX = None


def _subpix_ok(weight):
    return SUBPIX_UPSAMPLE and CONV_PRECISION != 'f32' and weight.shape[0] % 128 == 0


def _subpix_packed(weight, flip, f16c8=False):
    lib = L.load()
    cout, cin = weight.shape[0], weight.shape[1]
    we = torch.empty(4 * cout, cin, 3, 3, dtype=torch.float32, device=weight.device)
    L.check(lib.eovae_subpix_expand(L.ptr(weight), L.ptr(we), cout, cin, L.stream()), 'eovae_subpix_expand')
    return _pack3x3(we, 4 * cout, cin, flip, f16c8)


def _subpix_packed_pair(weight):
    """(forward, data-gradient) f16c8 packs of the four parity kernels from ONE expansion and one read of it."""
    lib = L.load()
    cout, cin = weight.shape[0], weight.shape[1]
    we = torch.empty(4 * cout, cin, 3, 3, dtype=torch.float32, device=weight.device)
    L.check(lib.eovae_subpix_expand(L.ptr(weight), L.ptr(we), cout, cin, L.stream()), 'eovae_subpix_expand')
    wp0 = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(4 * cout, cin, 0), dtype=torch.uint8, device=weight.device)
    wp1 = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(4 * cout, cin, 1), dtype=torch.uint8, device=weight.device)
    L.check(lib.eovae_conv3x3_f16c8_pack_pair(L.ptr(we), L.ptr(wp0), L.ptr(wp1), 4 * cout, cin, 1.0, L.stream()),
            'eovae_conv3x3_f16c8_pack_pair')
    return wp0, wp1


DOWN_DGRAD_UP_MX = os.environ.get('EOVAE_DOWN_DGRAD_UP_MX', '1') != '0'   # stride-2 data gradient through conv_f16c8_q16up.hip (0: 8-wave)


def _down_ok(x, weight):
    cout, cin = weight.shape[0], weight.shape[1]
    return (SUBPIX_UPSAMPLE and CONV_PRECISION != 'f32' and cin % 128 == 0 and x.shape[2] % 2 == 0
            and x.shape[3] % 2 == 0)


def _down_packed(weight, dgrad, f16c8=False):
    lib = L.load()
    cout, cin = weight.shape[0], weight.shape[1]
    we = torch.empty(4 * cout * cin * 9, dtype=torch.float32, device=weight.device)
    L.check(lib.eovae_down_expand(L.ptr(weight), L.ptr(we), cout, cin, int(dgrad), L.stream()),
            'eovae_down_expand')
    rows, cols = (4 * cin, cout) if dgrad else (cout, 4 * cin)   # as a forward layer [rows][cols][3][3]
    return _pack3x3(we, rows, cols, 0, f16c8)


def down_conv_s2d(x, weight, bias):
    """pad(0,1,0,1) + 3x3 stride 2 in space-to-depth form (K = 4*Cin gathered at the chunk parity)."""
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    f8 = _use_f16c8(cin)
    wp = _down_packed(weight, False, f8)
    y = torch.empty(n, cout, h // 2, w // 2, dtype=torch.float32, device=x.device)
    global _LAST_GN_SUMS
    sums = _gn_sums_buffer(n, cout, x.device) if f8 else None
    _launch3x3(x, wp, bias, None, y, n, cin, h, w, cout, h // 2, w // 2, MODE_NORMAL, 3, f8,
               range_amax(x) if f8 else None, gn_sums=sums)
    _LAST_GN_SUMS = sums
    return y


def down_conv_s2d_dgrad(dy, weight, h, w):
    n, cout = dy.shape[0], dy.shape[1]
    cin = weight.shape[1]
    f8 = _use_f16c8(cout)
    wp = _down_packed(weight, True, f8)
    dx = torch.empty(n, cin, h, w, dtype=torch.float32, device=dy.device)
    # Round 6: the stride-2 layer's data gradient IS an x2-upsample forward on the space-to-depth weights Wd'' -- parity (py, px) of
    # dx reads dy through the taps {py, py + 1} x {px, px + 1} of Wd'' (csrc/conv_bf16x3.hip down_expand_kernel: its non-zero taps
    # lie inside that window), which is conv_f16c8_q16up.hip's tap window exactly.  The kernel multiplies the window's zero taps
    # too (16 tap groups for 9), and still beats the 8-wave kernel's pixel-shuffle mode where it is taken (DOWN_DGRAD_UP_MX).
    lib = L.load()
    # (K = the layer's Cout >= 256: at 128 channels the tile is its epilogue -- 0.392 against 0.391 ms, profiles/r06k_down_layer_ab.log)
    if f8 and DOWN_DGRAD_UP_MX and cout >= 256 and lib.eovae_conv3x3_f16c8_up_mx_eligible(n, cout, h // 2, w // 2, cin) == 1:
        amax = amax_of(dy)
        rec = torch.empty_like(dy)
        L.check(lib.eovae_records_mx_from_f32(L.ptr(dy), L.ptr(amax), L.ptr(rec), n, cout, (h // 2) * (w // 2), L.stream()),
                'eovae_records_mx_from_f32')
        oa = zeros_small(1, torch.int32, dy.device) if EPILOGUE_AMAX else None
        L.check(lib.eovae_conv3x3_f16c8_up_mx(L.ptr(rec), L.ptr(wp), None, L.ptr(dx), L.ptr(amax), n, cout, h // 2, w // 2, cin,
                                              None, 0, L.ptr(oa), L.stream()), 'eovae_conv3x3_f16c8_up_mx')
        if oa is not None:
            _tag_amax(dx, oa)      # read by the data gradient of the block in front of this layer
        return dx
    oa = zeros_small(1, torch.int32, dy.device) if (f8 and EPILOGUE_AMAX) else None
    _launch3x3(dy, wp, None, None, dx, n, cout, h // 2, w // 2, cin, h, w, MODE_NORMAL, 4, f8,
               amax_of(dy) if f8 else None, out_amax=oa)
    if oa is not None:
        _tag_amax(dx, oa)      # read by the data gradient of the block in front of this layer
    return dx


def down_wgrad_rows(n, cin, h, w, cout):
    """True when down_conv_s2d_wgrad takes the row-streaming form for this layer (x is [n, cin, h, w]); that form does
    not produce the bias gradient (the caller leaves db to eovae_bias_grad)."""
    return bool(CONV_PRECISION == 'f16c8' and UP_WGRAD_ROWS and h % 2 == 0 and w % 2 == 0
                and L.load().eovae_conv3x3_f16c8_wgrad_down_eligible(n, cin, h // 2, w // 2, cout))


def down_conv_s2d_wgrad(x, dy, weight, db=None, xamax=None):
    """db: pre-zeroed [Cout], filled by the tile kernel's space-to-depth mode only (see down_wgrad_rows)."""
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    lib = L.load()
    if down_wgrad_rows(n, cin, h, w, cout):
        if xamax is None:
            xamax = range_amax(x)
        timed = WGRAD3X3_TIMER is not None
        if timed:
            ev0, ev1 = WGRAD3X3_TIMER.bracket(2.0 * n * (h // 2) * (w // 2) * cout * cin * 9, 4.0 * (x.numel() + dy.numel()))
            ev0.record()
        ws = workspace(lib.eovae_conv3x3_f16c8_wgrad_down_workspace(n, cin, h // 2, w // 2, cout), x.device)
        dw = torch.empty_like(weight)
        L.check(lib.eovae_conv3x3_f16c8_wgrad_down(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(amax_of(dy)), L.ptr(xamax), n, cin,
                                                   h // 2, w // 2, cout, L.ptr(ws), L.stream()), 'eovae_conv3x3_f16c8_wgrad_down')
        if timed:
            ev1.record()
        return dw
    dwp = torch.zeros(9 * cout * 4 * cin, dtype=torch.float32, device=x.device)
    _launch_wgrad3x3(x, dy, dwp, db, n, cin, h, w, cout, h // 2, w // 2, MODE_NORMAL, 2, xamax=xamax)
    dw = torch.empty_like(weight)
    L.check(lib.eovae_down_fold(L.ptr(dwp), L.ptr(dw), cout, cin, L.stream()), 'eovae_down_fold')
    return dw


def up_conv_subpix(x, weight, bias, want_bwd=False):
    """want_bwd (f16c8): also returns the data-gradient pack made from the same expansion (else None)."""
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    f8 = _use_f16c8(cin)
    wp_bwd = None
    if want_bwd and f8 and _use_f16c8(cout):
        wp, wp_bwd = _subpix_packed_pair(weight)
    else:
        wp = _subpix_packed(weight, False, f8)
    y = torch.empty(n, cout, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    # round 6: MX operand records of the low-res x (one HBM pass) + the DMA-fed 16 x 16 kernel that owns both column parities
    # (csrc/conv_f16c8_q16up.hip); kind 3 in the bench's timer
    up_mx = bool(f8 and L.load().eovae_conv3x3_f16c8_up_mx_eligible(n, cin, h, w, cout) == 1)
    timed = CONV3X3_TIMER is not None
    if timed:
        ev0, ev1 = CONV3X3_TIMER.bracket(2.0 * n * 4 * h * w * cout * cin * 9,
                                         4.0 * (x.numel() + y.numel() + weight.numel()), 3 if up_mx else 0)
        ev0.record()
    global _LAST_GN_SUMS
    sums = _gn_sums_buffer(n, cout, x.device) if f8 else None
    if up_mx:
        lib = L.load()
        amax = range_amax(x)
        rec = torch.empty_like(x)
        L.check(lib.eovae_records_mx_from_f32(L.ptr(x), L.ptr(amax), L.ptr(rec), n, cin, h * w, L.stream()),
                'eovae_records_mx_from_f32')
        L.check(lib.eovae_conv3x3_f16c8_up_mx(L.ptr(rec), L.ptr(wp), L.ptr(bias), L.ptr(y), L.ptr(amax), n, cin, h, w, cout,
                                              L.ptr(sums), GN_GROUPS_HINT if sums is not None else 0, None, L.stream()),
                'eovae_conv3x3_f16c8_up_mx')
    else:
        _launch3x3(x, wp, bias, None, y, n, cin, h, w, cout, 2 * h, 2 * w, MODE_NORMAL, 1, f8,
                   range_amax(x) if f8 else None, gn_sums=sums)
    _LAST_GN_SUMS = sums
    if timed:
        ev1.record()
    return (y, wp_bwd) if want_bwd else y


def up_conv_subpix_dgrad(dy, weight, h, w, wp=None):
    n, cout = dy.shape[0], dy.shape[1]
    cin = weight.shape[1]
    f8 = _use_f16c8(cout)
    if wp is None:
        wp = _subpix_packed(weight, True, f8)
    dx = torch.empty(n, cin, h, w, dtype=torch.float32, device=dy.device)
    # round 6: MX operand records of the high-res dY (one HBM pass) + the DMA-fed 16 x 16 kernel whose K runs over dY's four
    # parity planes (csrc/conv_f16c8_q16upg.hip); kind 3 in the bench's timer
    lib = L.load()
    up_mx = bool(f8 and lib.eovae_conv3x3_f16c8_up_dgrad_mx_eligible(n, cin, h, w, cout) == 1)
    timed = CONV3X3_TIMER is not None
    if timed:
        ev0, ev1 = CONV3X3_TIMER.bracket(2.0 * n * 4 * h * w * cout * cin * 9,
                                         4.0 * (dy.numel() + dx.numel() + weight.numel()), 3 if up_mx else 0)
        ev0.record()
    if up_mx:
        amax = amax_of(dy)
        rec = torch.empty_like(dy)
        L.check(lib.eovae_records_mx_from_f32(L.ptr(dy), L.ptr(amax), L.ptr(rec), n, cout, 4 * h * w, L.stream()),
                'eovae_records_mx_from_f32')
        L.check(lib.eovae_conv3x3_f16c8_up_dgrad_mx(L.ptr(rec), L.ptr(wp), L.ptr(dx), L.ptr(amax), n, cin, h, w, cout, L.stream()),
                'eovae_conv3x3_f16c8_up_dgrad_mx')
        if timed:
            ev1.record()
        return dx
    oa = zeros_small(1, torch.int32, dy.device) if (f8 and EPILOGUE_AMAX) else None
    _launch3x3(dy, wp, None, None, dx, n, cout, 2 * h, 2 * w, cin, h, w, MODE_NORMAL, 2, f8,
               amax_of(dy) if f8 else None, out_amax=oa)
    if oa is not None:
        _tag_amax(dx, oa)
    if timed:
        ev1.record()
    return dx


def up_conv_subpix_wgrad(x, dy, weight, db=None, xamax=None):
    """db (f16c8 mode only, pre-zeroed [Cout]): the bias gradient rides on the dy tiles the kernel stages."""
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    lib = L.load()
    if CONV_PRECISION == 'f16c8' and UP_WGRAD_ROWS and lib.eovae_conv3x3_f16c8_wgrad_up_eligible(n, cin, h, w, cout):
        # row-streaming sub-pixel form: records of x once, four parity passes in one launch, a deterministic fold
        if xamax is None:
            xamax = range_amax(x)
        timed = WGRAD3X3_TIMER is not None
        if timed:
            ev0, ev1 = WGRAD3X3_TIMER.bracket(2.0 * n * 4 * h * w * cout * cin * 9, 4.0 * (x.numel() + dy.numel()))
            ev0.record()
        ws = workspace(lib.eovae_conv3x3_f16c8_wgrad_up_workspace(n, cin, h, w, cout), x.device)
        dw = torch.empty_like(weight)
        L.check(lib.eovae_conv3x3_f16c8_wgrad_up(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(amax_of(dy)), L.ptr(xamax),
                                                 n, cin, h, w, cout, L.ptr(ws), L.stream()), 'eovae_conv3x3_f16c8_wgrad_up')
        if timed:
            ev1.record()
        return dw
    dwp = torch.zeros(9 * 4 * cout * cin, dtype=torch.float32, device=x.device)
    _launch_wgrad3x3(x, dy, dwp, db, n, cin, h, w, cout, 2 * h, 2 * w, MODE_NORMAL, 1, xamax=xamax)
    dw = torch.empty_like(weight)
    L.check(lib.eovae_subpix_fold(L.ptr(dwp), L.ptr(dw), cout, cin, L.stream()), 'eovae_subpix_fold')
    return dw


def _gemm_prec():
    return 0 if CONV_PRECISION == 'f32' else _PREC[CONV_PRECISION]


def conv1x1_lowp(x, weight, bias, residual):
    """y[n] = W x[n] (+bias, +residual) as a batched split-bf16 GEMM."""
    n, cin, h, w = x.shape
    cout, hw = weight.shape[0], h * w
    y = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    L.check(L.load().eovae_gemm_bf16x3(L.ptr(weight), L.ptr(x), L.ptr(y), L.ptr(bias), L.ptr(residual), n,
                                       cout, hw, cin, cin, 1, hw, 1, hw, 1, 0, cin * hw, cout * hw, 1.0, 1,
                                       0, _PREC[CONV_PRECISION], 1, 0, 0, L.stream()), 'eovae_gemm_bf16x3')
    return y


def conv1x1_lowp_dgrad(dy, weight):
    n, cout, h, w = dy.shape
    cin, hw = weight.shape[1], h * w
    dx = torch.empty(n, cin, h, w, dtype=torch.float32, device=dy.device)
    L.check(L.load().eovae_gemm_bf16x3(L.ptr(weight), L.ptr(dy), L.ptr(dx), None, None, n, cin, hw, cout,
                                       1, cin, hw, 1, hw, 1, 0, cout * hw, cin * hw, 1.0, 1, 0,
                                       _PREC[CONV_PRECISION], 1, 0, 0, L.stream()), 'eovae_gemm_bf16x3')
    return dx


def conv1x1_lowp_wgrad(x, dy):
    """dW[co][ci] = sum_n sum_p dy[n][co][p] x[n][ci][p]: batch- and K-split blocks add atomically."""
    n, cin, h, w = x.shape
    cout, hw = dy.shape[1], h * w
    dw = torch.zeros(cout, cin, 1, 1, dtype=torch.float32, device=x.device)
    # K = (image, pixel): the n image slabs are walked inside the kernel; K is cut into ~256 blocks
    tiles = ((cout + 127) // 128) * ((cin + 255) // 256)
    stages = n * ((hw + 31) // 32)
    ksplit = max(1, min(256 // max(1, tiles), stages // 4 if stages >= 8 else 1))
    L.check(L.load().eovae_gemm_bf16x3(L.ptr(dy), L.ptr(x), L.ptr(dw), None, None, 1, cout, cin, hw, hw, 1,
                                       1, hw, cin, 1, 0, 0, 0, 1.0, ksplit, 1, _PREC[CONV_PRECISION],
                                       n, cout * hw, cin * hw, L.stream()), 'eovae_gemm_bf16x3')
    return dw


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
        res = _c(residual) if residual is not None else None
        bias_c = _c(bias) if bias is not None else None
        packed = packed_format(x)
        if packed and not (ks == 3 and m == MODE_NORMAL and CONV_PRECISION == 'f16c8'):
            raise L.EovaeError('conv2d: operand-record input (GroupNorm packed output) is only understood by the '
                               "plain 3x3 f16c8 convolution")
        if packed == 2 and L.load().eovae_records_mx_ok(x.shape[0], cin, x.shape[2], x.shape[3], cout) != 1:
            # MX operand records are written for ONE consumer (the `cout` hint of group_norm_silu): a layer they were not
            # meant for would pass the forward and fail in its weight gradient, or -- on the thin kernels -- be read as format 1
            raise L.EovaeError(f'conv2d: MX operand records (format 2) handed to a layer that does not take them '
                               f'(Cin {cin}, Cout {cout}, {x.shape[2]}x{x.shape[3]}): the GroupNorm in front was told another '
                               'consumer (group_norm_silu(..., cout=...)) or a library option changed since')
        ctx.packed = packed
        ctx.wp_bwd = None
        if ks == 3 and m == MODE_UP and res is None and _subpix_ok(weight):
            if ctx.needs_input_grad[0]:     # (y, data-gradient pack from the same expansion)
                y, ctx.wp_bwd = up_conv_subpix(x, weight, bias_c, want_bwd=True)
            else:
                y = up_conv_subpix(x, weight, bias_c)
        elif ks == 3 and m == MODE_DOWN and res is None and _down_ok(x, weight):
            y = down_conv_s2d(x, weight, bias_c)
        elif (THIN_WGRAD and ks == 3 and m == MODE_NORMAL and res is None and CONV_PRECISION != 'f32' and
              L.load().eovae_conv3x3_thin_fwd_eligible(x.shape[0], cin, x.shape[2], x.shape[3], cout) == 1):
            # the wavelength-conditioned LAST layer (<= 4 output channels): an HBM-bound fp32 pass (csrc/conv_thin.hip)
            # instead of a 128-row matrix-core tile with 3 valid rows
            y = torch.empty(x.shape[0], cout, x.shape[2], x.shape[3], dtype=torch.float32, device=x.device)
            L.check(L.load().eovae_conv3x3_thin_fwd(L.ptr(x), L.ptr(weight), L.ptr(bias_c), L.ptr(y), x.shape[0], cin,
                                                    x.shape[2], x.shape[3], cout, int(packed), L.stream()),
                    'eovae_conv3x3_thin_fwd')
        elif ks == 3 and m != MODE_DOWN and CONV_PRECISION != 'f32':
            wp = None
            ctx.wp_bwd = None
            # both weight orientations from one amax pass when the data gradient of this layer will be asked for
            if m == MODE_NORMAL and _use_f16c8(cin) and _use_f16c8(cout) and ctx.needs_input_grad[0]:
                wp, ctx.wp_bwd = pack_pair_f16c8(weight)
            y = conv3x3_lowp(x, weight, bias_c, res, cout, m, False, packed=packed, wp=wp)
        elif ks == 1 and CONV_PRECISION != 'f32':
            y = conv1x1_lowp(x, weight, bias_c, res)
        else:
            wp, kp, mp = pack_weights(weight, m, False)
            y = conv_raw(x, wp, kp, mp, bias_c, res, cout, ks, m)
        ctx.save_for_backward(x, weight)
        # the forward's absmax of an un-normalised input serves the weight gradient too (the tag lives on the Python
        # object, which saved_tensors does not keep)
        tag = getattr(x, '_eovae_amax', None)
        ctx.x_amax = tag[0] if (tag is not None and tag[1] == x._version and not packed) else None
        ctx.mode, ctx.has_bias, ctx.has_res = m, bias is not None, residual is not None
        ctx.prec = CONV_PRECISION
        return y

    @staticmethod
    def backward(ctx, dy):
        with precision_scope(ctx.prec):   # same arithmetic as the forward, whatever the global says now
            return Conv2dFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _c(dy)
        m = ctx.mode
        cout, cin, ks, _ = weight.shape
        n, _, h, w = x.shape
        lib = L.load()
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        subpix = ks == 3 and m == MODE_UP and not ctx.has_res and _subpix_ok(weight)
        s2d = ks == 3 and m == MODE_DOWN and not ctx.has_res and _down_ok(x, weight)
        if ctx.needs_input_grad[0] and subpix:
            dx = up_conv_subpix_dgrad(dy, weight, h, w, wp=getattr(ctx, 'wp_bwd', None))
        elif ctx.needs_input_grad[0] and s2d:
            dx = down_conv_s2d_dgrad(dy, weight, h, w)
        elif ctx.needs_input_grad[0] and ks == 3 and CONV_PRECISION != 'f32':
            if m == MODE_NORMAL:
                dx = conv3x3_lowp(dy, weight, None, None, cin, MODE_NORMAL, True, wp=getattr(ctx, 'wp_bwd', None))
            elif m == MODE_DOWN:
                dx = conv3x3_lowp(dy, weight, None, None, cin, MODE_ZINS, True)
                if dx.shape[2] != h or dx.shape[3] != w:
                    raise L.EovaeError('conv2d(down): odd spatial sizes are not supported')
            else:
                du = conv3x3_lowp(dy, weight, None, None, cin, MODE_NORMAL, True)
                dx = torch.empty_like(x)
                L.check(lib.eovae_upsample2x_bwd(L.ptr(du), L.ptr(dx), n, cin, h, w, L.stream()),
                        'eovae_upsample2x_bwd')
        elif ctx.needs_input_grad[0] and ks == 1 and CONV_PRECISION != 'f32':
            dx = conv1x1_lowp_dgrad(dy, weight)
        elif ctx.needs_input_grad[0]:
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
        thin = (THIN_WGRAD and ks == 3 and m == MODE_NORMAL and CONV_PRECISION != 'f32' and ctx.needs_input_grad[1] and
                lib.eovae_conv3x3_thin_wgrad_eligible(n, cin, h, w, cout, int(ctx.packed)) == 1)
        if thin:
            # the wavelength-conditioned first / last layer with <= 4 channels on its thin side: an HBM-bound fp32 pass
            # (csrc/conv_thin.hip) instead of a 128-row matrix-core tile with 3 valid rows
            dw = torch.empty_like(weight)
            if want_db and cout > cin:
                db = torch.empty(cout, dtype=torch.float32, device=x.device)
            ws = workspace(lib.eovae_conv3x3_thin_wgrad_workspace(n, cin, h, w, cout), x.device)
            L.check(lib.eovae_conv3x3_thin_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), n, cin, h, w, cout,
                                                 int(ctx.packed), L.ptr(ws), L.stream()), 'eovae_conv3x3_thin_wgrad')
        elif ctx.needs_input_grad[1] and subpix:
            if want_db and CONV_PRECISION == 'f16c8':
                db = zeros_small(cout, torch.float32, x.device)
            dw = up_conv_subpix_wgrad(x, dy, weight, db, getattr(ctx, 'x_amax', None))
        elif ctx.needs_input_grad[1] and s2d:
            if want_db and not down_wgrad_rows(n, cin, h, w, cout):   # (the row-streaming form leaves db to eovae_bias_grad)
                db = zeros_small(cout, torch.float32, x.device)
            dw = down_conv_s2d_wgrad(x, dy, weight, db, getattr(ctx, 'x_amax', None))
        elif ctx.needs_input_grad[1] and ks == 1 and CONV_PRECISION != 'f32':
            dw = conv1x1_lowp_wgrad(x, dy)
        elif (ctx.needs_input_grad[1] and WGRAD_DIRECT and ks == 3 and m == MODE_NORMAL and CONV_PRECISION == 'f16c8'
              and cin % 32 == 0):
            dw = torch.empty_like(weight)
            if want_db:
                db = zeros_small(cout, torch.float32, x.device)
            _launch_wgrad3x3(x, dy, None, db, n, cin, h, w, cout, dy.shape[2], dy.shape[3], m, 0, packed=ctx.packed,
                             xamax=getattr(ctx, 'x_amax', None), dw=dw)
        elif ctx.needs_input_grad[1]:
            dwp = torch.zeros(ks * ks * cout * cin, dtype=torch.float32, device=x.device)
            if ks == 3 and m != MODE_DOWN and CONV_PRECISION != 'f32':
                if want_db:  # bias gradient rides on the dy tiles the wgrad kernel stages anyway
                    db = zeros_small(cout, torch.float32, x.device)
                _launch_wgrad3x3(x, dy, dwp, db, n, cin, h, w, cout, dy.shape[2], dy.shape[3], m, 0,
                                 packed=ctx.packed, xamax=getattr(ctx, 'x_amax', None))
            else:
                L.check(lib.eovae_conv2d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), n, cin, h, w, cout,
                                               dy.shape[2], dy.shape[3], ks, m, L.stream()),
                        'eovae_conv2d_wgrad')
            dw = torch.empty_like(weight)
            L.check(lib.eovae_conv_unpack_wgrad(L.ptr(dwp), L.ptr(dw), cout, cin, ks, L.stream()),
                    'eovae_conv_unpack_wgrad')
        if want_db and db is None:
            db = torch.empty(cout, dtype=torch.float32, device=x.device)
            L.check(lib.eovae_bias_grad(L.ptr(dy), L.ptr(db), n, cout, dy.shape[2] * dy.shape[3],
                                        L.stream()), 'eovae_bias_grad')
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None


def conv2d(x, weight, bias=None, residual=None, mode='normal'):
    global _LAST_GN_SUMS, _LAST_OUT_AMAX
    _LAST_GN_SUMS = None
    _LAST_OUT_AMAX = None
    y = Conv2dFn.apply(x, weight, bias, residual, mode)
    if _LAST_GN_SUMS is not None:
        y._eovae_gn_sums = (_LAST_GN_SUMS, GN_GROUPS_HINT, y._version)
        _LAST_GN_SUMS = None
    if _LAST_OUT_AMAX is not None:
        _tag_amax(y, _LAST_OUT_AMAX)
        _LAST_OUT_AMAX = None
    return y


# ------------------------------------------------------------------------------------------
# GroupNorm (+ swish)
# ------------------------------------------------------------------------------------------


def _gn_forward(x, gamma, beta, groups, eps, silu, packed):
    """packed: 0 fp32 result, 1 operand records, 2 MX operand records."""
    n, c, h, w = x.shape
    lib = L.load()
    ws = workspace(lib.eovae_gn_workspace_bytes(n, c, h * w, groups), x.device)
    y = torch.empty_like(x)
    stats = torch.empty(n * groups * 2, dtype=torch.float32, device=x.device)
    fn = (lib.eovae_gn_silu_fwd, lib.eovae_gn_silu_fwd_packed, lib.eovae_gn_silu_fwd_packed_mx)[int(packed)]
    L.check(fn(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(stats), L.ptr(gn_sums_of(x, groups)), n, c,
               h * w, groups, eps, int(silu), L.ptr(ws), L.stream()), 'eovae_gn_silu_fwd')
    return y, stats


def _mark(y, packed):
    return mark_packed_mx(y) if packed == 2 else (mark_packed(y) if packed else y)


class GroupNormSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, packed=False):
        L.require_gpu(x, 'group_norm')
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        n, c, h, w = x.shape
        lib = L.load()
        y, stats = _gn_forward(x, gamma, beta, groups, eps, silu, packed)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.groups, ctx.silu = groups, silu
        return _mark(y, packed)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        dy = _c(dy)
        n, c, h, w = x.shape
        lib = L.load()
        ws = workspace(lib.eovae_gn_workspace_bytes(n, c, h * w, ctx.groups), x.device)
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        amax = _new_amax(dx)
        L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                      None, L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(amax), n, c, h * w,
                                      ctx.groups, int(ctx.silu), L.ptr(ws), L.stream()),
                'eovae_gn_silu_bwd')
        _tag_amax(dx, amax)
        return dx, dgamma, dbeta, None, None, None, None


def gn_timeout_report():
    """What the timed-out wave saw (csrc/groupnorm.hip gnf_timed_out), as text for the error message."""
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    if L.load().eovae_gn_bwd_timeout_info(buf) != 0:
        return 'no diagnostics'
    tok, t0, d1, d2, clk, restarts, shape = (int(buf[i]) for i in range(7))
    return (f'token {tok:#x}; wait window started at tick {t0}, elapsed readings {d1} and {d2} ticks of 10 ns (bound 2e8), '
            f'shader clock counter {clk}, window restarts {restarts}; call N={shape >> 48} C={(shape >> 32) & 0xffff} '
            f'HW={shape & 0xffffffff}')


# Data-parallel runs: a time-out is seen by ONE rank; raising there alone leaves the others in the next bucket all-reduce
# until RCCL's watchdog fires.  parallel.BucketedGradReducer reduces a fault flag with every step's gradients
# and installs this hook: it returns True on EVERY rank in the same step once any rank has reported a time-out.
GN_TIMEOUT_AGREED = None    # the latest reducer's fault_agreed (kept for callers that look at it; the check walks GN_TIMEOUT_HOOKS)
GN_TIMEOUT_HOOKS: list = []  # weakref.WeakMethod of every live BucketedGradReducer.fault_agreed (one global slot was
                             # overwritten by a second reducer and kept a dropped reducer's bound method alive)


def raise_on_gn_backward_timeouts():
    """Once per step (training_step): a team of the one-pass GroupNorm backward that timed out has poisoned its
    dgamma / dbeta with NaN; the kernels also leave the event in a host-visible word.  Raises instead of letting the
    optimiser write NaN into every weight.  Sees every launch the host has synchronised with -- at the
    latest one step after the event (data parallel: three steps, and on every rank in the same step)."""
    lib = L.load()
    hooks = [h for h in (r() for r in GN_TIMEOUT_HOOKS) if h is not None]   # live reducers only (weak references)
    if hooks:                                # data parallel: the local word stays set until the ranks agree
        if not any(h() for h in hooks):
            return
        tok = lib.eovae_gn_bwd_timeouts(0)
        what = gn_timeout_report() if tok else 'reported by another rank'
        lib.eovae_gn_bwd_timeouts(1)
        if os.environ.get('EOVAE_GN_TIMEOUT_ACTION') == 'warn':      # diagnosis only: every rank sees the same reduced flag, so
            print(f'[eo_vae_amd] GroupNorm backward time-out agreed by the ranks ({what})', flush=True)   # every rank prints and goes on
            return
        raise L.EovaeError('GroupNorm backward: a workgroup team timed out waiting for its partial sums on some rank '
                           f'({what}); the gradients of that step are NaN -- set EOVAE_GN_BWD_FUSED=0 to use the two-pass kernels')
    tok = lib.eovae_gn_bwd_timeouts(0)
    if not tok:
        return
    what = gn_timeout_report()
    lib.eovae_gn_bwd_timeouts(1)
    if os.environ.get('EOVAE_GN_TIMEOUT_ACTION') == 'warn':      # diagnosis only
        print(f'[eo_vae_amd] GroupNorm backward time-out recorded ({what})', flush=True)
        return
    raise L.EovaeError(f'GroupNorm backward: a workgroup team timed out waiting for its partial sums ({what}); '
                       'the gradients of that step are NaN -- set EOVAE_GN_BWD_FUSED=0 to use the two-pass kernels')


def group_norm_silu(x, gamma, beta, groups=32, eps=1e-6, silu=True, packed=False, cout=None):
    """packed=True: the result is handed over as f16c8 operand records (see mark_packed) -- only for a
    consumer that is a plain 3x3 Conv2dFn in f16c8 mode.  cout: that layer's output channels, if the caller knows them --
    the records are then MX operand records where the layer takes them (records_format_for)."""
    packed = int(bool(packed and silu and packed_ok(x.shape[1])))
    if packed:
        packed = records_format_for(x, cout)
    y = GroupNormSiluFn.apply(x, gamma, beta, groups, eps, silu, packed)
    return _mark(y, packed)


class GroupNormSiluForkFn(torch.autograd.Function):
    """(y, x_skip) = (silu(gn(x)), x): the residual branch of a ResnetBlock leaves through this
    node so that its gradient is folded into the GroupNorm backward kernel (dx = gn_bwd + dskip)
    instead of a separate elementwise-add pass over the activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, packed=False):
        L.require_gpu(x, 'group_norm')
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        y, stats = _gn_forward(x, gamma, beta, groups, eps, silu, packed)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.groups, ctx.silu = groups, silu
        return _mark(y, packed), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, gamma, beta, stats = ctx.saved_tensors
        dy = _c(dy)
        dskip = _c(dskip) if dskip is not None else None
        n, c, h, w = x.shape
        lib = L.load()
        ws = workspace(lib.eovae_gn_workspace_bytes(n, c, h * w, ctx.groups), x.device)
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        amax = _new_amax(dx)
        L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                      L.ptr(dskip), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(amax), n, c,
                                      h * w, ctx.groups, int(ctx.silu), L.ptr(ws), L.stream()),
                'eovae_gn_silu_bwd')
        _tag_amax(dx, amax)
        return dx, dgamma, dbeta, None, None, None, None


def group_norm_silu_fork(x, gamma, beta, groups=32, eps=1e-6, silu=True, packed=False, cout=None):
    packed = int(bool(packed and silu and packed_ok(x.shape[1])))
    if packed:
        packed = records_format_for(x, cout)
    y, skip = GroupNormSiluForkFn.apply(x, gamma, beta, groups, eps, silu, packed)
    return _mark(y, packed), skip


# ------------------------------------------------------------------------------------------
# attention core (softmax(Q^T K / sqrt(C)) V on [B, C, T] tensors)
# ------------------------------------------------------------------------------------------


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        L.require_gpu(q, 'attention')
        q, k, v = _c(q), _c(k), _c(v)
        b, c, t = q.shape
        p = torch.empty(b, t, t, dtype=torch.float32, device=q.device)
        o = torch.empty_like(q)
        L.check(L.load().eovae_attention_fwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(p), L.ptr(o), b, c, t,
                                             scale, _gemm_prec(), L.stream()), 'eovae_attention_fwd')
        ctx.save_for_backward(q, k, v, p)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        do = _c(do)
        b, c, t = q.shape
        dp = torch.empty_like(p)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        L.check(L.load().eovae_attention_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(p), L.ptr(do),
                                             L.ptr(dp), L.ptr(dq), L.ptr(dk), L.ptr(dv), b, c, t,
                                             ctx.scale, _gemm_prec(), L.stream()), 'eovae_attention_bwd')
        return dq, dk, dv, None


def attention(q, k, v, scale):
    return AttentionFn.apply(q, k, v, scale)


# ------------------------------------------------------------------------------------------
# posterior, latent normalisation
# ------------------------------------------------------------------------------------------


class PosteriorSampleFn(torch.autograd.Function):
    """z = mean + std * eps (eps=None -> mode)."""

    @staticmethod
    def forward(ctx, moments, eps):
        L.require_gpu(moments, 'posterior')
        moments = _c(moments)
        b, c2, h, w = moments.shape
        z = torch.empty(b, c2 // 2, h, w, dtype=torch.float32, device=moments.device)
        if eps is not None:
            # the kernel reads eps with the latent's extents: a smaller tensor is an out-of-bounds read on the device (a test
            # of round 6 faulted the GPU that way), and the reference's mean + std * eps would raise a broadcast error
            if tuple(eps.shape) != (b, c2 // 2, h, w) or eps.dtype != torch.float32 or eps.device != moments.device:
                raise L.EovaeError(f'posterior sample: noise is {tuple(eps.shape)} {eps.dtype} on {eps.device}, the latent is '
                                   f'{(b, c2 // 2, h, w)} float32 on {moments.device}')
        eps = _c(eps) if eps is not None else None
        L.check(L.load().eovae_posterior_fwd(L.ptr(moments), L.ptr(eps), L.ptr(z), None, b, c2 // 2,
                                             h * w, L.stream()), 'eovae_posterior_fwd')
        ctx.save_for_backward(moments, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        moments, eps = ctx.saved_tensors
        b, c2, h, w = moments.shape
        dm = torch.empty_like(moments)
        L.check(L.load().eovae_posterior_bwd(L.ptr(moments), L.ptr(eps), L.ptr(_c(dz)), None,
                                             L.ptr(dm), b, c2 // 2, h * w, L.stream()),
                'eovae_posterior_bwd')
        return dm, None


class PosteriorKLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, moments):
        L.require_gpu(moments, 'posterior.kl')
        moments = _c(moments)
        b, c2, h, w = moments.shape
        kl = torch.empty(b, dtype=torch.float32, device=moments.device)
        L.check(L.load().eovae_posterior_fwd(L.ptr(moments), None, None, L.ptr(kl), b, c2 // 2, h * w,
                                             L.stream()), 'eovae_posterior_fwd')
        ctx.save_for_backward(moments)
        return kl

    @staticmethod
    def backward(ctx, dkl):
        (moments,) = ctx.saved_tensors
        b, c2, h, w = moments.shape
        dm = torch.empty_like(moments)
        L.check(L.load().eovae_posterior_bwd(L.ptr(moments), None, None, L.ptr(_c(dkl)), L.ptr(dm), b,
                                             c2 // 2, h * w, L.stream()), 'eovae_posterior_bwd')
        return dm


class LatentNormFn(torch.autograd.Function):
    """spatial z [B,Z,H,W] -> packed, batch-normalised zp [B,4Z,H/2,W/2] (running stats updated
    in training mode, exactly as nn.BatchNorm2d(affine=False) does)."""

    @staticmethod
    def forward(ctx, z, run_mean, run_var, training, eps, momentum):
        L.require_gpu(z, 'latent_norm')
        z = _c(z)
        b, zc, h, w = z.shape
        zp = torch.empty(b, 4 * zc, h // 2, w // 2, dtype=torch.float32, device=z.device)
        save = torch.empty(4 * zc * 2, dtype=torch.float32, device=z.device)
        L.check(L.load().eovae_latent_bn_fwd(L.ptr(z), L.ptr(zp), L.ptr(run_mean), L.ptr(run_var),
                                             L.ptr(save), b, zc, h, w, int(training), eps, momentum,
                                             L.stream()), 'eovae_latent_bn_fwd')
        ctx.save_for_backward(zp, save)
        ctx.training, ctx.shape = training, (b, zc, h, w)
        return zp

    @staticmethod
    def backward(ctx, dzp):
        zp, save = ctx.saved_tensors
        b, zc, h, w = ctx.shape
        dz = torch.empty(b, zc, h, w, dtype=torch.float32, device=zp.device)
        L.check(L.load().eovae_latent_bn_bwd(L.ptr(zp), L.ptr(_c(dzp)), L.ptr(save), L.ptr(dz), b, zc,
                                             h, w, int(ctx.training), L.stream()),
                'eovae_latent_bn_bwd')
        return dz, None, None, None, None, None


class LatentInvNormFn(torch.autograd.Function):
    """packed normalised zp [B,4Z,h,w] -> spatial z [B,Z,2h,2w] = zp*sqrt(rv+eps)+rm."""

    @staticmethod
    def forward(ctx, zp, run_mean, run_var, eps):
        L.require_gpu(zp, 'latent_inv_norm')
        zp = _c(zp)
        b, c4, h2, w2 = zp.shape
        z = torch.empty(b, c4 // 4, 2 * h2, 2 * w2, dtype=torch.float32, device=zp.device)
        L.check(L.load().eovae_latent_inv(L.ptr(zp), L.ptr(z), L.ptr(run_mean), L.ptr(run_var), b,
                                          c4 // 4, 2 * h2, 2 * w2, eps, 0, L.stream()),
                'eovae_latent_inv')
        # the reference reads the running stats at forward time; keep a private copy for backward
        ctx.save_for_backward(run_mean.clone(), run_var.clone())
        ctx.eps, ctx.shape = eps, (b, c4 // 4, 2 * h2, 2 * w2)
        return z

    @staticmethod
    def backward(ctx, dz):
        rm, rv = ctx.saved_tensors
        b, zc, h, w = ctx.shape
        dzp = torch.empty(b, 4 * zc, h // 2, w // 2, dtype=torch.float32, device=dz.device)
        L.check(L.load().eovae_latent_inv(L.ptr(_c(dz)), L.ptr(dzp), L.ptr(rm), L.ptr(rv), b, zc, h, w,
                                          ctx.eps, 1, L.stream()), 'eovae_latent_inv')
        return dzp, None, None, None


# ------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------


class PixelLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind, eps):
        L.require_gpu(pred, 'pixel_loss')
        pred, target = _c(pred), _c(target)
        if pred.shape != target.shape:
            raise L.EovaeError(f'pixel_loss: shapes differ {tuple(pred.shape)} vs {tuple(target.shape)}')
        lib = L.load()
        ws = workspace(lib.eovae_loss_workspace_bytes(), pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        L.check(lib.eovae_pixel_loss_fwd(L.ptr(pred), L.ptr(target), pred.numel(), kind, eps,
                                         L.ptr(out), L.ptr(ws), L.stream()), 'eovae_pixel_loss_fwd')
        ctx.save_for_backward(pred, target)
        ctx.kind, ctx.eps = kind, eps
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        L.check(L.load().eovae_pixel_loss_bwd(L.ptr(pred), L.ptr(target), L.ptr(_c(gout)), L.ptr(dpred),
                                              pred.numel(), ctx.kind, ctx.eps, L.stream()),
                'eovae_pixel_loss_bwd')
        return dpred, None, None, None


def l1_loss(pred, target):
    return PixelLossFn.apply(pred, target, 0, 0.0)


def charbonnier_loss(pred, target, eps=1e-3):
    return PixelLossFn.apply(pred, target, 1, eps)


def mse_loss(pred, target):
    return PixelLossFn.apply(pred, target, 2, 0.0)


class SAMLossFn(torch.autograd.Function):
    """mean(1 - cos(pred, target)) along the channel axis (consistency_loss.py:186-209)."""

    @staticmethod
    def forward(ctx, pred, target, eps):
        L.require_gpu(pred, 'sam_loss')
        pred, target = _c(pred), _c(target)
        if pred.shape != target.shape or pred.dim() != 4:
            raise L.EovaeError(f'sam_loss: shapes {tuple(pred.shape)} vs {tuple(target.shape)}')
        n, c, h, w = pred.shape
        lib = L.load()
        ws = workspace(lib.eovae_loss_workspace_bytes(), pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        L.check(lib.eovae_sam_loss_fwd(L.ptr(pred), L.ptr(target), n, c, h * w, eps, L.ptr(out), L.ptr(ws),
                                       L.stream()), 'eovae_sam_loss_fwd')
        ctx.save_for_backward(pred, target)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        n, c, h, w = pred.shape
        dpred = torch.empty_like(pred)
        L.check(L.load().eovae_sam_loss_bwd(L.ptr(pred), L.ptr(target), L.ptr(_c(gout)), L.ptr(dpred), n, c, h * w,
                                            ctx.eps, L.stream()), 'eovae_sam_loss_bwd')
        return dpred, None, None


def sam_loss(pred, target, eps=1e-8):
    return SAMLossFn.apply(pred, target, eps)


class GradientDifferenceLossFn(torch.autograd.Function):
    """GradientDifferenceLoss(alpha=1) (consistency_loss.py:243-269)."""

    @staticmethod
    def forward(ctx, pred, target):
        L.require_gpu(pred, 'gdl_loss')
        pred, target = _c(pred), _c(target)
        if pred.shape != target.shape or pred.dim() != 4:
            raise L.EovaeError(f'gdl_loss: shapes {tuple(pred.shape)} vs {tuple(target.shape)}')
        n, c, h, w = pred.shape
        lib = L.load()
        ws = workspace(lib.eovae_loss_workspace_bytes(), pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        L.check(lib.eovae_gdl_loss_fwd(L.ptr(pred), L.ptr(target), n * c, h, w, L.ptr(out), L.ptr(ws), L.stream()),
                'eovae_gdl_loss_fwd')
        ctx.save_for_backward(pred, target)
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        n, c, h, w = pred.shape
        dpred = torch.empty_like(pred)
        L.check(L.load().eovae_gdl_loss_bwd(L.ptr(pred), L.ptr(target), L.ptr(_c(gout)), L.ptr(dpred), n * c, h, w,
                                            L.stream()), 'eovae_gdl_loss_bwd')
        return dpred, None


def gradient_difference_loss(pred, target):
    return GradientDifferenceLossFn.apply(pred, target)


class FocalFrequencyLossFn(torch.autograd.Function):
    """FocalFrequencyLoss (ffl.py) with ave_spectrum=False, batch_matrix=True; the DFTs are fp32 MFMA GEMMs."""

    @staticmethod
    def forward(ctx, pred, target, alpha, patch_factor, log_matrix):
        L.require_gpu(pred, 'focal_frequency_loss')
        pred, target = _c(pred.float()), _c(target.float())
        if pred.shape != target.shape or pred.dim() != 4:
            raise L.EovaeError(f'focal_frequency_loss: shapes {tuple(pred.shape)} vs {tuple(target.shape)}')
        b, c, h, w = pred.shape
        lib = L.load()
        nbytes = lib.eovae_ffl_workspace_bytes(b, c, h, w, patch_factor)
        if nbytes == 0:
            raise L.EovaeError(f'focal_frequency_loss: patch_factor {patch_factor} does not fit {h}x{w}')
        ws = workspace(nbytes, pred.device)
        gy = torch.empty(lib.eovae_ffl_gy_elems(b, c, h, w, patch_factor), dtype=torch.float32, device=pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        L.check(lib.eovae_ffl_fwd(L.ptr(pred), L.ptr(target), L.ptr(out), L.ptr(gy), L.ptr(ws), b, c, h, w,
                                  patch_factor, alpha, int(log_matrix), L.stream()), 'eovae_ffl_fwd')
        ctx.save_for_backward(gy)
        ctx.geo = (b, c, h, w, patch_factor)
        return out

    @staticmethod
    def backward(ctx, gout):
        gy, = ctx.saved_tensors
        b, c, h, w, pf = ctx.geo
        lib = L.load()
        ws = workspace(lib.eovae_ffl_workspace_bytes(b, c, h, w, pf), gy.device)
        dpred = torch.empty((b, c, h, w), dtype=torch.float32, device=gy.device)
        L.check(lib.eovae_ffl_bwd(L.ptr(gy), L.ptr(_c(gout)), L.ptr(dpred), L.ptr(ws), b, c, h, w, pf, L.stream()),
                'eovae_ffl_bwd')
        return dpred, None, None, None, None


def focal_frequency_loss(pred, target, alpha=1.0, patch_factor=2, log_matrix=True):
    return FocalFrequencyLossFn.apply(pred, target, float(alpha), int(patch_factor), bool(log_matrix))


# ------------------------------------------------------------------------------------------
# EQ-VAE latent transforms (new_autoencoder.py:456-460, 520-531, 612-627)
# ------------------------------------------------------------------------------------------


def _quarter_turns(angle):
    """rot90(k, dims=[-1,-2]) of the reference -> quarter turns in the dims=[-2,-1] sense of the kernels."""
    return 0 if angle is None else (4 - int(angle)) % 4


class ResizeRotFn(torch.autograd.Function):
    """rot90(F.interpolate(z, size, 'bilinear', align_corners=False), angle, dims=[-1,-2]) in one gather
    kernel (the batch-preparation kernel without z-score/clamp); backward = its adjoint."""

    @staticmethod
    def forward(ctx, z, ht, wt, angle):
        L.require_gpu(z, 'resize_rot')
        z = _c(z)
        n, c, hs, ws_ = z.shape
        rot = _quarter_turns(angle)
        hd, wd = (wt, ht) if rot & 1 else (ht, wt)
        out = torch.empty((n, c, hd, wd), dtype=torch.float32, device=z.device)
        L.check(L.load().eovae_batch_prep(L.ptr(z), None, None, L.ptr(out), n * c, 1, hs, ws_, ht, wt, 1.0, 0.0,
                                          0, 0, rot, L.stream()), 'eovae_batch_prep')
        ctx.geo = (n, c, hs, ws_, ht, wt, rot)
        return out

    @staticmethod
    def backward(ctx, dy):
        n, c, hs, ws_, ht, wt, rot = ctx.geo
        dy = _c(dy)
        dz = torch.empty((n, c, hs, ws_), dtype=torch.float32, device=dy.device)
        L.check(L.load().eovae_resize_rot_bwd(L.ptr(dy), L.ptr(dz), n * c, hs, ws_, ht, wt, rot, L.stream()),
                'eovae_resize_rot_bwd')
        return dz, None, None, None


def resize_rot(z, size, angle=None):
    return ResizeRotFn.apply(z, int(size[0]), int(size[1]), angle)


@torch.no_grad()
def area_rot(images, size, angle=None):
    """rot90(F.interpolate(images, size, mode='area'), angle, dims=[-1,-2]); `size` is the size BEFORE the
    rotation, exactly as the reference passes recon.shape[-2:] (new_autoencoder.py:612-617)."""
    L.require_gpu(images, 'area_rot')
    images = _c(images)
    n, c, hs, ws_ = images.shape
    ht, wt = int(size[0]), int(size[1])
    rot = _quarter_turns(angle)
    hd, wd = (wt, ht) if rot & 1 else (ht, wt)
    out = torch.empty((n, c, hd, wd), dtype=torch.float32, device=images.device)
    L.check(L.load().eovae_area_rot(L.ptr(images), L.ptr(out), n * c, hs, ws_, ht, wt, rot, L.stream()),
            'eovae_area_rot')
    return out


# ------------------------------------------------------------------------------------------
# optimiser primitives on flat buffers
# ------------------------------------------------------------------------------------------


def grad_norm(flat_grad: torch.Tensor, max_norm: float, out: torch.Tensor) -> torch.Tensor:
    """out[0] = ||g||, out[1] = clip coefficient (device side, no host sync)."""
    lib = L.load()
    ws = workspace(lib.eovae_gradnorm_workspace_bytes(), flat_grad.device)
    L.check(lib.eovae_grad_norm(L.ptr(flat_grad), flat_grad.numel(), float(max_norm or 0.0),
                                L.ptr(out), L.ptr(ws), L.stream()), 'eovae_grad_norm')
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, clip=None, grad_scale=1.0, weight_decay=0.0):
    L.check(L.load().eovae_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, beta1,
                                     beta2, eps, step, L.ptr(clip), grad_scale, weight_decay, L.stream()),
            'eovae_adam_step')


def scale_inplace(g, clip=None, s=1.0):
    L.check(L.load().eovae_scale_inplace(L.ptr(g), g.numel(), L.ptr(clip), s, L.stream()),
            'eovae_scale_inplace')


# ------------------------------------------------------------------------------------------
# MS-SSIM (5-scale Gaussian pyramid) and encode_latents running statistics
# ------------------------------------------------------------------------------------------


class MSSSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, data_range, betas):
        L.require_gpu(pred, 'msssim')
        pred, target = _c(pred), _c(target)
        b, c, h, w = pred.shape
        lib = L.load()
        nbytes = lib.eovae_msssim_workspace_bytes(b, c, h, w, len(betas))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)  # kept until backward
        bt = torch.tensor(betas, dtype=torch.float32, device=pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        L.check(lib.eovae_msssim_fwd(L.ptr(pred), L.ptr(target), L.ptr(out), b, c, h, w, len(betas),
                                     L.ptr(bt), data_range, L.ptr(ws), L.stream()), 'eovae_msssim_fwd')
        ctx.save_for_backward(pred, target, ws, bt)
        ctx.data_range = data_range
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target, ws, bt = ctx.saved_tensors
        b, c, h, w = pred.shape
        dpred = torch.empty_like(pred)
        L.check(L.load().eovae_msssim_bwd(L.ptr(pred), L.ptr(target), L.ptr(_c(gout)), L.ptr(dpred), b, c,
                                          h, w, bt.numel(), L.ptr(bt), ctx.data_range, L.ptr(ws),
                                          L.stream()), 'eovae_msssim_bwd')
        return dpred, None, None, None


def msssim(pred, target, data_range=6.0, betas=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333)):
    return MSSSIMFn.apply(pred, target, float(data_range), tuple(betas))


def running_stats_update(z: torch.Tensor, state: torch.Tensor) -> None:
    """state [5, C] = (mean, var, min, max, count); z [B, C, H, W]."""
    L.require_gpu(z, 'running_stats')
    z = _c(z)
    b, c = z.shape[:2]
    L.check(L.load().eovae_running_stats_update(L.ptr(z), L.ptr(state), b, c, z[0, 0].numel(), L.stream()),
            'eovae_running_stats_update')
