"""ctypes binding of the C-ABI in include/eovae_hip.h (libeovae_hip.so).

The HIP library is the product path: there is NO fallback.  Missing library, missing GPU
or a non-zero status from any entry point raises -- loudly."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EOVAE_LIB') or os.path.join(_HERE, 'libeovae_hip.so')

_p, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)
_ll = C.c_longlong

# name -> (restype, argtypes); mirrors include/eovae_hip.h one to one
SIGNATURES = {
    'eovae_last_error': (C.c_char_p, []),
    'eovae_abi_version': (_i, []),
    'eovae_target_arch': (C.c_char_p, []),
    'eovae_build_flags': (_i, []),
    'eovae_clock_probe': (_i, [_p, _p, _i, _i, _p]),
    'eovae_set_option': (_i, [C.c_char_p, _i]),
    'eovae_get_option': (_i, [C.c_char_p]),
    'eovae_conv_packed_dims': (_i, [_i, _i, _i, _i, _i, _ip, _ip]),
    'eovae_conv_pack_weights': (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p]),
    'eovae_conv2d_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    'eovae_conv3x3_bf16_packed_elems': (_sz, [_i, _i, _i]),
    'eovae_conv3x3_bf16_pack': (_i, [_p, _p, _i, _i, _i, _f, _p]),
    'eovae_conv3x3_bf16_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_absmax_bits': (_i, [_p, _sz, _p, _p]),
    'eovae_conv3x3_f16c8_packed_bytes': (_sz, [_i, _i, _i]),
    'eovae_conv3x3_f16c8_pack': (_i, [_p, _p, _i, _i, _i, _f, _p]),
    'eovae_conv3x3_f16c8_pack_pair': (_i, [_p, _p, _p, _i, _i, _f, _p]),
    'eovae_conv3x3_f16c8_pack_multi': (_i, [_p, _i, _f, _p]),
    'eovae_conv3x3_f16c8_fwd': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    'eovae_conv3x3_f16c8_fwd_amax': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    'eovae_conv3x3_f16c8_fwd_mx': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    'eovae_conv3x3_f16c8_which': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'eovae_conv3x3_f16c8_up_mx_eligible': (_i, [_i, _i, _i, _i, _i]),
    'eovae_records_mx_from_f32': (_i, [_p, _p, _p, _i, _i, _i, _p]),
    'eovae_conv3x3_f16c8_up_mx': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    'eovae_conv3x3_f16c8_up_dgrad_mx_eligible': (_i, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_f16c8_up_dgrad_mx': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'eovae_conv3x3_f16c8_wgrad': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_conv3x3_f16c8_wgrad_dw_workspace': (_sz, [_i, _i]),
    'eovae_conv3x3_f16c8_wgrad_dw': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    'eovae_gn_silu_fwd_packed': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p]),
    'eovae_gn_silu_fwd_packed_mx': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p]),
    'eovae_records_mx_ok': (_i, [_i, _i, _i, _i, _i]),
    'eovae_subpix_expand': (_i, [_p, _p, _i, _i, _p]),
    'eovae_down_expand': (_i, [_p, _p, _i, _i, _i, _p]),
    'eovae_down_fold': (_i, [_p, _p, _i, _i, _p]),
    'eovae_subpix_fold': (_i, [_p, _p, _i, _i, _p]),
    'eovae_conv3x3_bf16_wgrad': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_conv2d_wgrad': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_conv_unpack_wgrad': (_i, [_p, _p, _i, _i, _i, _p]),
    'eovae_bias_grad': (_i, [_p, _p, _i, _i, _i, _p]),
    'eovae_upsample2x_bwd': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'eovae_gn_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'eovae_gn_silu_fwd': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p]),
    'eovae_gn_silu_bwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'eovae_conv3x3_thin_wgrad_workspace': (_sz, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_thin_wgrad_eligible': (_i, [_i, _i, _i, _i, _i, _i]),
    'eovae_conv3x3_thin_wgrad': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    'eovae_conv3x3_thin_fwd_eligible': (_i, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_thin_fwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_gn_bwd_fused': (_i, [_i]),
    'eovae_gn_bwd_timeouts': (C.c_ulonglong, [_i]),
    'eovae_gn_bwd_timeout_info': (_i, [C.POINTER(C.c_ulonglong)]),
    'eovae_gn_bwd_fused_eligible': (_i, [_i, _i, _i, _i]),
    'eovae_gemm_f32': (_i, [_p, _p, _p, _i, _i, _i, _i] + [_ll] * 9 + [_f, _p]),
    'eovae_softmax_rows': (_i, [_p, _ll, _i, _p]),
    'eovae_softmax_bwd_rows': (_i, [_p, _p, _ll, _i, _p]),
    'eovae_conv3x3_f16c8_wgrad_up_eligible': (_i, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_f16c8_wgrad_up_workspace': (_sz, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_f16c8_wgrad_up': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'eovae_conv3x3_f16c8_wgrad_down_eligible': (_i, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_f16c8_wgrad_down_workspace': (_sz, [_i, _i, _i, _i, _i]),
    'eovae_conv3x3_f16c8_wgrad_down': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'eovae_amax_bound_from_sums': (_i, [_p, _i, _p, _p]),
    'eovae_gen_gemm': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i] + [_ll] * 8 + [_f, _i, _i, _ll, _p]),
    'eovae_gen_ln_fwd': (_i, [_p, _i, _ll, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    'eovae_gen_ln_bwd': (_i, [_p, _p, _i, _ll, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    'eovae_gen_colsum': (_i, [_p, _p, _i, _i, _ll, _p]),
    'eovae_gemm_bf16x3': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i] + [_ll] * 9 + [_f, _i, _i, _i, _i, _ll, _ll, _p]),
    'eovae_attention_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    'eovae_attention_bwd': (_i, [_p] * 9 + [_i, _i, _i, _f, _i, _p]),
    'eovae_posterior_fwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    'eovae_posterior_bwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'eovae_latent_bn_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p]),
    'eovae_latent_bn_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'eovae_latent_inv': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    'eovae_loss_workspace_bytes': (_sz, []),
    'eovae_pixel_loss_fwd': (_i, [_p, _p, _sz, _i, _f, _p, _p, _p]),
    'eovae_pixel_loss_bwd': (_i, [_p, _p, _p, _p, _sz, _i, _f, _p]),
    'eovae_sam_loss_fwd': (_i, [_p, _p, _i, _i, _i, _f, _p, _p, _p]),
    'eovae_sam_loss_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p]),
    'eovae_gdl_loss_fwd': (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    'eovae_gdl_loss_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    'eovae_ffl_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'eovae_ffl_gy_elems': (_sz, [_i, _i, _i, _i, _i]),
    'eovae_ffl_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    'eovae_ffl_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'eovae_msssim_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'eovae_msssim_fwd': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p, _p]),
    'eovae_msssim_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p, _p]),
    'eovae_running_stats_update': (_i, [_p, _p, _i, _i, _i, _p]),
    'eovae_batch_prep': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _i, _p]),
    'eovae_resize_rot_bwd': (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_area_rot': (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'eovae_gradnorm_workspace_bytes': (_sz, []),
    'eovae_grad_norm': (_i, [_p, _sz, _f, _p, _p, _p]),
    'eovae_adam_step': (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _i, _p, _f, _f, _p]),
    'eovae_scale_inplace': (_i, [_p, _sz, _p, _f, _p]),
    'eovae_multi_gather': (_i, [_p, _i, _p, _p]),
}

_lib = None
ABI_VERSION = 12  # include/eovae_hip.h; bumped whenever a declared signature changes


class EovaeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libeovae_hip.so (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EovaeError(
            f'{LIB_PATH} is missing: build it with `python -m eo_vae_amd.build` '
            '(eo_vae_amd has no CPU/PyTorch fallback for its kernels)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.eovae_abi_version() != ABI_VERSION:
        raise EovaeError(f'{LIB_PATH} has C-ABI version {lib.eovae_abi_version()}, this package binds '
                         f'version {ABI_VERSION}: rebuild with `python -m eo_vae_amd.build`')
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().eovae_last_error().decode(errors='replace')
        raise EovaeError(f'{what} failed (status {rc}): {msg}')


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise EovaeError(f'{what}: tensor is on {t.device}; eo_vae_amd kernels only run on an '
                         'AMD GPU (no CPU fallback) -- move the model and inputs to "cuda"')
    if t.dtype != torch.float32:
        raise EovaeError(f'{what}: dtype {t.dtype} unsupported (fp32 storage)')


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
