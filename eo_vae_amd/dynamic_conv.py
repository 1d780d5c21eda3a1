"""Wavelength-conditioned first/last convolutions (reference: modules/dynamic_conv.py).

The tiny hyper-network (sin-cos embedding -> FCResLayer -> 4-layer post-norm transformer over
[128 weight tokens | C wave tokens | 1 bias token], < 0.3 GFLOP): its encoder layers and Linear
layers run on the small exact-fp32 kernels of csrc/generator.hip (generator_ops.py; round 4), the element-wise glue
around them on PyTorch-ROCm ops; the convolution with the generated kernel and its three gradients run on the HIP
implicit-GEMM / thin-layer kernels."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init
from torch.nn.attention import SDPBackend, sdpa_kernel

from . import generator_ops, ops


def get_1d_sincos_pos_embed_from_grid_torch(embed_dim: int, pos: torch.Tensor) -> torch.Tensor:
    """[M] positions -> [M, D] = [sin(pos*w) | cos(pos*w)], w_d = 10000^(-d/(D/2)) (:37-59)."""
    assert embed_dim % 2 == 0
    half = embed_dim // 2
    omega = torch.arange(half, dtype=torch.float32, device=pos.device) / float(half)
    omega = 1.0 / 10000 ** omega
    ang = pos.reshape(-1)[:, None] * omega[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


# The embedding of the wavelengths is a pure function of `wvs` (no parameters): ~11 tiny launches at the head of every
# step for a [C, 256] matrix that only changes with the modality.  Kept per tensor OBJECT (strong reference, so the id cannot
# be recycled) and `_version` (in-place edits), as the generator cache below; never for a tensor that requires grad.
_SINCOS_CACHE: dict = {}


def wavelength_embedding(embed_dim: int, wvs: torch.Tensor) -> torch.Tensor:
    """get_1d_sincos_pos_embed_from_grid_torch(embed_dim, wvs * 1000) (dynamic_conv.py:466-468, 640-642), memoised."""
    if wvs.requires_grad or torch.is_inference_mode_enabled():
        return get_1d_sincos_pos_embed_from_grid_torch(embed_dim, wvs * 1000)
    key = (id(wvs), embed_dim)
    hit = _SINCOS_CACHE.get(key)
    if hit is not None and hit[0] is wvs and hit[1] == wvs._version:
        return hit[2]
    with torch.no_grad():
        emb = get_1d_sincos_pos_embed_from_grid_torch(embed_dim, wvs * 1000)
    if len(_SINCOS_CACHE) >= 64:
        _SINCOS_CACHE.clear()
    _SINCOS_CACHE[key] = (wvs, wvs._version, emb)
    return emb


class FCResLayer(nn.Module):  # :336-366
    def __init__(self, linear_size: int = 128):
        super().__init__()
        self.l_size = linear_size
        self.w1 = nn.Linear(linear_size, linear_size)
        self.w2 = nn.Linear(linear_size, linear_size)

    def forward(self, x):
        if GENERATOR_KERNELS:
            lin = generator_ops.linear
            return x + F.relu(lin(F.relu(lin(x, self.w1.weight, self.w1.bias)), self.w2.weight, self.w2.bias))
        return x + F.relu(self.w2(F.relu(self.w1(x))))


# The encoder's hyper-network is the head (forward, before conv_in) and the tail (backward, after the last weight
# gradient) of every train step: ~110 + ~190 five-microsecond launches = 3.6 ms that nothing can overlap (DESIGN.md
# s7).  `nn.TransformerEncoderLayer` + the MATH SDPA path spend ~24 launches per layer on reshapes, scalings and
# copies; the same post-norm layer written out on the same parameters needs 13 (one in_proj GEMM, a scaled
# batched QK^T, softmax, PV, one head-merge copy, out_proj, add, LayerNorm, FFN GEMM + GELU + GEMM, add, LayerNorm).
FAST_GENERATOR = os.environ.get('EOVAE_FAST_GENERATOR', '1') != '0'
# Round 4: the layer and the Linear layers around it are HIP kernels of this product (csrc/generator.hip through
# generator_ops: 9 launches per layer forward, ~22 backward, no BLAS library).  EOVAE_GENERATOR_KERNELS=0 keeps the
# PyTorch-op formulation below for A/B timing and as the fp32 reference of tests/test_gpu_generator.py.
GENERATOR_KERNELS = os.environ.get('EOVAE_GENERATOR_KERNELS', '1') != '0'
DEAD_ROW_ELIMINATION = os.environ.get('EOVAE_GENERATOR_DEAD_ROWS', '1') != '0'   # 0: the last layer computes all rows

# The hyper-network's GEMMs are 132-row problems ([132, 256] x [256, 256 .. 2048]).  PyTorch-ROCm's default BLAS
# (hipBLASLt) picks 256-row macro tiles for them: 19-40 us per call where rocBLAS needs 7-10 (tools/gen_kernels.py).
# The BLAS preference is PROCESS-WIDE state, so importing this module does not touch it: the entry points
# that own the process -- trainer.Trainer, train.py, bench.py -- call prefer_rocblas_for_generator(); a host
# application that embeds the model decides for itself.  EOVAE_GENERATOR_BLAS=keep makes the call a no-op.
_BLAS_STATE = {'backend': None}


def prefer_rocblas_for_generator():
    """Switch PyTorch's GEMM backend to rocBLAS (process-wide; every other GEMM of this product is its own HIP kernel).
    Returns the name of the backend in force afterwards; logs it once."""
    if _BLAS_STATE['backend'] is not None:
        return _BLAS_STATE['backend']
    if GENERATOR_KERNELS:    # round 4: the generators run on this product's own kernels -- no process-wide switch to make
        _BLAS_STATE['backend'] = 'none (own HIP kernels, csrc/generator.hip)'
        return _BLAS_STATE['backend']
    backend = 'default'
    if os.environ.get('EOVAE_GENERATOR_BLAS', 'rocblas') == 'rocblas':
        if torch.cuda.is_available() and torch.cuda.is_initialized() and os.environ.get('DISABLE_ADDMM_CUDA_LT') != '1':
            # PyTorch reads DISABLE_ADDMM_CUDA_LT at its first addmm: called this late, the bias-fused GEMMs may stay on
            # hipBLASLt (2-4 x slower per call at 132 rows, results differ by up to ~2e-4) -- said out loud, not guessed
            import warnings
            warnings.warn('prefer_rocblas_for_generator() called after CUDA was initialised: set DISABLE_ADDMM_CUDA_LT=1 in '
                          'the environment (or call it before the first GEMM) for every hyper-network GEMM to use rocBLAS')
        os.environ.setdefault('DISABLE_ADDMM_CUDA_LT', '1')       # addmm + bias would still go to hipBLASLt (read once by
        try:                                                      # PyTorch: effective when set before the first addmm)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                torch.backends.cuda.preferred_blas_library('cublas')   # = rocBLAS on ROCm
            backend = str(torch.backends.cuda.preferred_blas_library()).rsplit('.', 1)[-1]
        except Exception as e:     # builds without the switch keep their default -- said out loud, not swallowed
            backend = f'default (switch unavailable: {type(e).__name__})'
    _BLAS_STATE['backend'] = backend
    print(f'[eo_vae_amd] hyper-network GEMM backend: {backend}', flush=True)
    return backend


def generator_blas_backend():
    """The backend prefer_rocblas_for_generator() left in force (None: never called in this process)."""
    return _BLAS_STATE['backend']


def _encoder_layer(layer, x, keep_from=0):
    """nn.TransformerEncoderLayer.forward for an unbatched [L, d] input, post-norm, exact GELU, dropout 0
    (the configuration of dynamic_conv.py:86-101), on the layer's own parameters.

    keep_from > 0 (LAST layer of a generator): only the output rows keep_from .. L - 1 are computed.  The generators read
    nothing but the wavelength rows and the bias row of the final output (dynamic_conv.py:121-130, 173-183), every op
    behind the attention scores is row-wise, so the other rows' queries, out_proj, LayerNorms and FFN are dead work
    (128 of 132 rows at C = 3: a quarter of a 4-layer generator, forward and backward); the kept rows see the same
    arithmetic (a library GEMM may block 4 rows differently from 132: last-bit differences)."""
    att = layer.self_attn
    seq_len, d = x.shape
    h = att.num_heads
    dh = d // h
    if keep_from > 0:
        wq, wkv = att.in_proj_weight[:d], att.in_proj_weight[d:]
        bq, bkv = att.in_proj_bias[:d], att.in_proj_bias[d:]
        xq = x[keep_from:]
        nq = seq_len - keep_from
        q = F.linear(xq, wq, bq).view(nq, h, dh).transpose(0, 1)                # [h, nq, dh]
        kv = F.linear(x, wkv, bkv).view(seq_len, 2, h, dh)
        k, v = kv[:, 0].transpose(0, 1), kv[:, 1].transpose(0, 1)              # [h, L, dh]
        s = torch.baddbmm(torch.empty(h, nq, seq_len, dtype=x.dtype, device=x.device), q, k.transpose(1, 2),
                          beta=0.0, alpha=float(dh) ** -0.5)
        o = torch.bmm(torch.softmax(s, dim=-1), v).transpose(0, 1).reshape(nq, d)
        x = xq
    else:
        qkv = F.linear(x, att.in_proj_weight, att.in_proj_bias).view(seq_len, 3, h, dh)
        q, k, v = (qkv[:, i].transpose(0, 1) for i in range(3))                  # [h, L, dh] strided views
        s = torch.baddbmm(torch.empty(h, seq_len, seq_len, dtype=x.dtype, device=x.device), q, k.transpose(1, 2),
                          beta=0.0, alpha=float(dh) ** -0.5)
        o = torch.bmm(torch.softmax(s, dim=-1), v).transpose(0, 1).reshape(seq_len, d)
    x = F.layer_norm(x + F.linear(o, att.out_proj.weight, att.out_proj.bias), (d,), layer.norm1.weight,
                     layer.norm1.bias, layer.norm1.eps)
    f = F.linear(F.gelu(F.linear(x, layer.linear1.weight, layer.linear1.bias)), layer.linear2.weight,
                 layer.linear2.bias)
    return F.layer_norm(x + f, (d,), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)


def _layer_is_plain(layer):
    return (isinstance(layer, nn.TransformerEncoderLayer) and not layer.norm_first
            and layer.activation_relu_or_gelu == 2 and layer.dropout.p == 0.0 and layer.dropout1.p == 0.0
            and layer.dropout2.p == 0.0 and layer.self_attn._qkv_same_embed_dim and layer.self_attn.bias_k is None
            and not layer.self_attn.add_zero_attn)


class TransformerWeightGenerator(nn.Module):  # :62-130
    _bias_out = None  # encoder flavour: one bias vector of embed_dim

    def __init__(self, input_dim, output_dim, embed_dim, num_heads=4, num_layers=1):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=input_dim, nhead=num_heads, activation='gelu',
                                           norm_first=False, batch_first=False, dropout=0.0)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)
        self.fc_weight = nn.Linear(input_dim, output_dim)
        self.fc_bias = nn.Linear(input_dim, self._bias_out or embed_dim)
        self.wt_num = 128
        self.weight_tokens = nn.Parameter(torch.empty(self.wt_num, input_dim))
        self.bias_token = nn.Parameter(torch.empty(1, input_dim))
        init.normal_(self.weight_tokens, std=0.02)
        init.normal_(self.bias_token, std=0.02)

    def _tokens(self, waves):
        seq = torch.cat([self.weight_tokens, waves, self.bias_token], dim=0)
        enc = self.transformer_encoder
        if FAST_GENERATOR and enc.norm is None and all(_layer_is_plain(lyr) for lyr in enc.layers):
            n = len(enc.layers)
            layer_fn = generator_ops.encoder_layer if GENERATOR_KERNELS else _encoder_layer
            for i, lyr in enumerate(enc.layers):   # the last layer only computes the rows the generator reads
                seq = layer_fn(lyr, seq, self.wt_num if (i == n - 1 and DEAD_ROW_ELIMINATION) else 0)
            if DEAD_ROW_ELIMINATION:
                return seq                          # rows wt_num .. L - 1
            return seq[self.wt_num:]
        return enc(seq)[self.wt_num:]  # unbatched [L, d]

    def _fc(self, lin, x):
        return generator_ops.linear(x, lin.weight, lin.bias) if GENERATOR_KERNELS else lin(x)

    def forward(self, waves):
        out = self._tokens(waves)                   # [C + 1, d]: the wavelength rows and the bias row
        return self._fc(self.fc_weight, out[:-1] + waves), self._fc(self.fc_bias, out[-1])


class TransformerWeightGenerator_decoder(TransformerWeightGenerator):  # :133-183
    _bias_out = 1

    def forward(self, waves):
        out = self._tokens(waves)
        wave_out = out[:-1]
        bias = self._fc(self.fc_bias, wave_out + self.bias_token.repeat(waves.shape[0], 1))
        return self._fc(self.fc_weight, wave_out + waves), bias


def _xavier_linear(m):
    if isinstance(m, nn.Linear):
        init.xavier_uniform_(m.weight)
        if m.bias is not None:
            m.bias.data.fill_(0.01)


class _DynamicBase(nn.Module):
    generator_cls = TransformerWeightGenerator

    def __init__(self, wv_planes, inter_dim=128, kernel_size=3, stride=1, padding=1, embed_dim=128,
                 num_layers=1, num_heads=4, generator_type='transformer', rank_ratio=4):
        super().__init__()
        if generator_type != 'transformer':
            raise NotImplementedError("only generator_type='transformer' is on the hot path "
                                      '(configs/eo-vae.yaml)')
        if kernel_size != 3 or stride != 1 or padding != 1:
            raise NotImplementedError('dynamic conv: 3x3, stride 1, padding 1 only')
        self.kernel_size, self.wv_planes, self.embed_dim = kernel_size, wv_planes, embed_dim
        self._num_kernel = kernel_size * kernel_size * embed_dim
        self.inter_dim, self.stride, self.padding = inter_dim, stride, padding
        self.generator_type, self.use_weight_standardization = generator_type, False
        self.weight_generator = self.generator_cls(wv_planes, self._num_kernel, embed_dim,
                                                   num_heads=num_heads, num_layers=num_layers)
        self.scaler = 0.1
        self.fclayer = FCResLayer(wv_planes)
        self.weight_generator.apply(_xavier_linear)
        self.fclayer.apply(_xavier_linear)
        self._gen_cache = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._gen_cache.clear())

    def _raw(self, wvs):
        """Generator output for `wvs`.  In inference (eval mode, no grad) the result depends only on the
        wavelength VALUES and the parameters, so it is cached ('cacheable per wvs in eval');
        train(), load_state_dict() and .to() drop the cache.

        Two levels.  (1) identity: the cache keeps a strong reference to every tensor object it has
        seen, so `id(wvs)` cannot be recycled while the entry lives, and `_version` catches in-place
        edits -- the same tensor passed again costs no host read.  (2) values: a tensor object not seen
        before is read back once (a [C]-element copy; the only stream synchronisation) and looked up by
        its values, then registered under (1).  A key made of data_ptr/_version alone (round 1) handed
        a recycled address the previous wavelengths' kernels."""
        cacheable = not self.training and not torch.is_grad_enabled()
        if cacheable:
            hit = self._gen_cache.get(('id', id(wvs)))
            if hit is not None and hit[0] is wvs and hit[1] == wvs._version:
                return hit[2]
            vkey = ('val', str(wvs.device), wvs.dtype, tuple(wvs.shape),
                    tuple(wvs.detach().reshape(-1).tolist()))
            out = self._gen_cache.get(vkey)
            if out is not None:
                self._remember(('id', id(wvs)), (wvs, wvs._version, out))
                return out
        waves = wavelength_embedding(self.wv_planes, wvs)
        # The 4-head, <= 142-token self-attention of the generator runs on the MATH path of SDPA (two batched
        # GEMMs + softmax): PyTorch's fused path on ROCm is aotriton, i.e. Triton-built kernels, which this
        # product does not put into its step (north_star: no Triton); at this size there is nothing to fuse.
        with sdpa_kernel(SDPBackend.MATH):
            out = self.weight_generator(self.fclayer(waves))
        if cacheable:
            self._remember(vkey, out)
            self._remember(('id', id(wvs)), (wvs, wvs._version, out))
        return out

    def _remember(self, key, value, limit=32):
        if len(self._gen_cache) >= limit:
            self._gen_cache.clear()
        self._gen_cache[key] = value

    def train(self, mode: bool = True):
        self._gen_cache.clear()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._gen_cache.clear()
        return super()._apply(fn, *args, **kwargs)


class DynamicConv(_DynamicBase):
    """C -> embed_dim, kernel [E, C, 3, 3] generated per wavelength (:369-535)."""

    def get_distillation_weight(self, wvs_microns):
        weight, bias = self._raw(wvs_microns)
        k = self.kernel_size
        w = weight.view(wvs_microns.size(0), k, k, self.embed_dim).permute(3, 0, 1, 2)
        return w * self.scaler, bias.view(self.embed_dim) * self.scaler

    def forward(self, img_feat, wvs):
        w, b = self.get_distillation_weight(wvs)
        return ops.conv2d(img_feat, w, b, None, 'normal')


class DynamicConv_decoder(_DynamicBase):
    """embed_dim -> C, kernel [C, E, 3, 3]; the bias is scaled by 0.1 twice in forward
    (:692-697) but once in get_distillation_weight (:659-664) -- reproduced as is."""
    generator_cls = TransformerWeightGenerator_decoder

    def __init__(self, wv_planes, inter_dim=128, kernel_size=3, stride=1, padding=1, embed_dim=128,
                 num_layers=2, num_heads=4, generator_type='transformer', rank_ratio=4):
        super().__init__(wv_planes, inter_dim, kernel_size, stride, padding, embed_dim, num_layers,
                         num_heads, generator_type, rank_ratio)

    def get_distillation_weight(self, wvs_microns):
        weight, bias = self._raw(wvs_microns)
        k = self.kernel_size
        w = weight.view(wvs_microns.size(0), k, k, self.embed_dim).permute(0, 3, 1, 2)
        return w * self.scaler, bias.squeeze() * self.scaler

    # The generator is ~50 tiny library GEMMs that leave most of the chip idle and whose result is needed
    # only by the LAST op of the decoder: Decoder.forward starts it on a side HIP stream before the body
    # (prefetch), and forward() joins the stream here.  Autograd replays each op's backward on the stream
    # of its forward, so the generator's backward overlaps the decoder body's backward the same way.
    _side_stream = None
    _pending = None

    def prefetch(self, waves):
        if not waves.is_cuda or os.environ.get('EOVAE_HYPERNET_STREAM', '1') == '0':
            return
        cls = DynamicConv_decoder
        if cls._side_stream is None:
            cls._side_stream = torch.cuda.Stream()
            from .parallel import SIDE_STREAMS
            SIDE_STREAMS.append(cls._side_stream)
        side, main = cls._side_stream, torch.cuda.current_stream()
        side.wait_stream(main)          # parameters / wavelengths written on the main stream
        with torch.cuda.stream(side):
            w, b = self.get_distillation_weight(waves)
            b = (b * self.scaler).reshape(-1)
        self._pending = (w, b)

    def forward(self, img_feat, waves):
        if self._pending is not None:
            w, b = self._pending
            self._pending = None
            main = torch.cuda.current_stream()
            main.wait_stream(DynamicConv_decoder._side_stream)
            w.record_stream(main), b.record_stream(main)
        else:
            w, b = self.get_distillation_weight(waves)
            b = (b * self.scaler).reshape(-1)
        self.weight = w  # read by EOFluxVAE.get_last_layer (:708)
        return ops.conv2d(img_feat, w, b, None, 'normal')
