"""CPU oracle for the EOFluxVAE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain PyTorch-CPU (fp32) restatement of the algorithm of the
reference's data-parallel hot path (nilsleh/eo-vae).  It exists so that the
hand-written HIP kernels in ``eo_vae_amd/csrc`` can be checked on the GPU box,
where the reference itself is not available.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``eo_vae_amd`` never does.

Pinning: every class/function below is compared against the *reference itself*
(imported from /root/reference in the build container) by
``tests/golden/gen_golden.py``, which also writes the committed fixtures under
``tests/golden/``.  One exception: MS-SSIM.  The reference delegates it to the
third-party ``torchmetrics`` package (not vendored, not pinned, not installed
here) -- ``msssim`` below restates torchmetrics' published algorithm and is
"PARITY UNPINNED" (see DESIGN.md).

Citations are ``file:line`` relative to /root/reference.
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# deterministic parameter fill (shared by golden generator and all tests)
# --------------------------------------------------------------------------


def _hash_uniform(n: int, salt: int) -> torch.Tensor:
    """Integer-hash pseudo-random numbers in [-0.5, 0.5); pure int64 torch ops
    so every platform regenerates identical values (no RNG state involved)."""
    i = torch.arange(n, dtype=torch.int64)
    h = (i * 2654435761 + salt * 40503 + 12345) & 0xFFFFFFFF
    h = h ^ (h >> 15)
    h = (h * 2246822519) & 0xFFFFFFFF
    h = h ^ (h >> 13)
    h = (h * 3266489917) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return (h & 0xFFFFFF).to(torch.float64) / float(1 << 24) - 0.5


@torch.no_grad()
def det_fill(module: nn.Module, seed: int = 0, gain: float = 1.0) -> None:
    """Closed-form parameter fill: no weight files, no RNG.

    >=2-D tensors: uniform with std = gain/sqrt(fan_in); norm weights 1+-0.1;
    1-D biases +-0.05; token parameters std 0.02.  BatchNorm running stats get
    a non-trivial deterministic state as well."""
    for idx, (name, p) in enumerate(module.state_dict().items()):
        if not torch.is_floating_point(p):
            continue
        u = _hash_uniform(p.numel(), seed * 1000003 + idx).reshape(p.shape)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'running_mean':
            v = 0.5 * u
        elif leaf == 'running_var':
            v = 1.0 + 0.8 * u
        elif 'token' in leaf:
            v = u * math.sqrt(12.0) * 0.02
        elif p.dim() >= 2:
            fan_in = p.numel() // p.shape[0]
            v = u * math.sqrt(12.0) * gain / math.sqrt(fan_in)
        elif 'norm' in name and leaf == 'weight':
            v = 1.0 + 0.2 * u
        else:
            v = 0.1 * u
        p.copy_(v.to(p.dtype))


def synthetic_batch(batch: int, channels: int, size: int, seed: int = 1234):
    """Synthetic z-score-like patches: low-res Gaussian field
    bilinearly upsampled x8 + 0.25 white noise, clamped to [-2.5, 8]."""
    g = torch.Generator('cpu').manual_seed(seed)
    low = torch.randn(batch, channels, max(size // 8, 1), max(size // 8, 1), generator=g)
    x = F.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
    x = x + 0.25 * torch.randn(batch, channels, size, size, generator=g)
    return x.clamp_(-2.5, 8.0)


WAVELENGTHS = {  # eo_vae/datasets/terramesh_datamodule.py:18-50
    'S2RGB': [0.665, 0.56, 0.49],
    'S1RTC': [5.4, 5.6],
    'S2L2A': [0.443, 0.490, 0.560, 0.665, 0.705, 0.740, 0.783, 0.842, 0.865, 1.610, 2.190, 0.945],
    'S2L1C': [0.443, 0.490, 0.560, 0.665, 0.705, 0.740, 0.783, 0.842, 0.865, 0.945, 1.375, 1.610, 2.190],
}

# --------------------------------------------------------------------------
# layers  (eo_vae/models/modules/layers.py)
# --------------------------------------------------------------------------


def swish(x):  # layers.py:21-22
    return x * torch.sigmoid(x)


def _gn(c):  # layers.py:14-18 -- 32 groups, eps 1e-6, affine
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class Downsample(nn.Module):  # layers.py:25-37
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))  # zero pad right/bottom only


class Upsample(nn.Module):  # layers.py:40-50
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=1, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class ResnetBlock(nn.Module):  # layers.py:53-114 (AdaIN branch inactive: cond_dim=None)
    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = _gn(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = _gn(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, emb=None):
        h = self.conv1(swish(self.norm1(x)))
        h = self.conv2(swish(self.norm2(h)))
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):  # layers.py:117-142 -- one head, d = C, scale C^-1/2
    def __init__(self, c):
        super().__init__()
        self.norm = _gn(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        hn = self.norm(x)
        q = self.q(hn).reshape(b, c, h * w)
        k = self.k(hn).reshape(b, c, h * w)
        v = self.v(hn).reshape(b, c, h * w)
        s = torch.einsum('bci,bcj->bij', q, k) * (c ** -0.5)
        p = torch.softmax(s, dim=-1)
        o = torch.einsum('bij,bcj->bci', p, v).reshape(b, c, h, w)
        return x + self.proj_out(o)


# --------------------------------------------------------------------------
# wavelength-conditioned convolutions  (eo_vae/models/modules/dynamic_conv.py)
# --------------------------------------------------------------------------


def sincos_embed(dim: int, pos: torch.Tensor) -> torch.Tensor:  # dynamic_conv.py:37-59
    omega = torch.arange(dim // 2, dtype=torch.float32, device=pos.device)
    omega = 1.0 / 10000 ** (omega / (dim / 2.0))
    out = pos.reshape(-1)[:, None] * omega[None, :]
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


class FCResLayer(nn.Module):  # dynamic_conv.py:336-366
    def __init__(self, d):
        super().__init__()
        self.w1 = nn.Linear(d, d)
        self.w2 = nn.Linear(d, d)

    def forward(self, x):
        return x + F.relu(self.w2(F.relu(self.w1(x))))


class TransformerWeightGenerator(nn.Module):  # dynamic_conv.py:62-130
    decoder_bias = False

    def __init__(self, d, out_dim, embed_dim, num_heads=4, num_layers=1):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=d, nhead=num_heads, activation='gelu',
                                           norm_first=False, batch_first=False, dropout=0.0)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)
        self.fc_weight = nn.Linear(d, out_dim)
        self.fc_bias = nn.Linear(d, 1 if self.decoder_bias else embed_dim)
        self.wt_num = 128
        self.weight_tokens = nn.Parameter(torch.zeros(self.wt_num, d))
        self.bias_token = nn.Parameter(torch.zeros(1, d))

    def forward(self, waves):
        x = torch.cat([self.weight_tokens, waves, self.bias_token], dim=0)
        out = self.transformer_encoder(x)  # unbatched [L, d]
        weights = self.fc_weight(out[self.wt_num:-1] + waves)
        if self.decoder_bias:  # dynamic_conv.py:162-183
            bias = self.fc_bias(out[self.wt_num:-1] + self.bias_token.repeat(waves.shape[0], 1))
        else:
            bias = self.fc_bias(out[-1])
        return weights, bias


class TransformerWeightGenerator_decoder(TransformerWeightGenerator):  # dynamic_conv.py:133-183
    decoder_bias = True


class DynamicConv(nn.Module):  # dynamic_conv.py:369-535 (transformer generator only)
    def __init__(self, wv_planes, inter_dim=128, kernel_size=3, stride=1, padding=1,
                 embed_dim=128, num_layers=1, num_heads=4):
        super().__init__()
        self.wv_planes, self.embed_dim, self.kernel_size = wv_planes, embed_dim, kernel_size
        self.stride, self.padding, self.scaler = stride, padding, 0.1
        self.weight_generator = TransformerWeightGenerator(
            wv_planes, kernel_size * kernel_size * embed_dim, embed_dim, num_heads, num_layers)
        self.fclayer = FCResLayer(wv_planes)

    def generate(self, wvs):
        waves = self.fclayer(sincos_embed(self.wv_planes, wvs * 1000))
        weight, bias = self.weight_generator(waves)
        k = self.kernel_size
        w = weight.view(wvs.numel(), k, k, self.embed_dim).permute(3, 0, 1, 2)  # [E,C,k,k]
        return w * self.scaler, bias.view(self.embed_dim) * self.scaler

    get_distillation_weight = generate  # dynamic_conv.py:471-497: same weight and bias as forward

    def forward(self, x, wvs):
        w, b = self.generate(wvs)
        return F.conv2d(x, w, b, stride=self.stride, padding=self.padding)


class DynamicConv_decoder(nn.Module):  # dynamic_conv.py:538-710
    def __init__(self, wv_planes, inter_dim=128, kernel_size=3, stride=1, padding=1,
                 embed_dim=128, num_layers=2, num_heads=4):
        super().__init__()
        self.wv_planes, self.embed_dim, self.kernel_size = wv_planes, embed_dim, kernel_size
        self.stride, self.padding, self.scaler = stride, padding, 0.1
        self.weight_generator = TransformerWeightGenerator_decoder(
            wv_planes, kernel_size * kernel_size * embed_dim, embed_dim, num_heads, num_layers)
        self.fclayer = FCResLayer(wv_planes)

    def generate(self, wvs):
        waves = self.fclayer(sincos_embed(self.wv_planes, wvs * 1000))
        weight, bias = self.weight_generator(waves)
        k = self.kernel_size
        w = weight.view(wvs.numel(), k, k, self.embed_dim).permute(0, 3, 1, 2)  # [C,E,k,k]
        # bias is scaled TWICE in forward (dynamic_conv.py:692-697)
        return w * self.scaler, bias.squeeze() * self.scaler * self.scaler

    def get_distillation_weight(self, wvs):  # dynamic_conv.py:638-664: the bias is scaled ONCE here
        w, b = self.generate(wvs)
        return w, b / self.scaler

    def forward(self, x, wvs):
        w, b = self.generate(wvs)
        self.weight = w  # dynamic_conv.py:708 (get_last_layer)
        return F.conv2d(x, w, b, stride=self.stride, padding=self.padding)


# --------------------------------------------------------------------------
# Encoder / Decoder  (eo_vae/models/model.py)
# --------------------------------------------------------------------------


def _dyn_kwargs(kw):
    kw = dict(kw or {})
    kw.pop('use_adain', None)
    kw.pop('mode', None)
    return kw.pop('wv_planes', 128), kw.pop('inter_dim', 128), kw


class Encoder(nn.Module):  # model.py:67-197
    def __init__(self, resolution, in_channels, ch, ch_mult, num_res_blocks, z_channels,
                 use_dynamic_ops=False, dynamic_conv_kwargs=None):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.z_channels = resolution, in_channels, z_channels
        self.use_dynamic_ops = use_dynamic_ops
        if use_dynamic_ops:
            wv, inter, kw = _dyn_kwargs(dynamic_conv_kwargs)
            self.conv_in = DynamicConv(wv_planes=wv, inter_dim=inter, embed_dim=ch, **kw)
        else:
            self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(self.num_resolutions):
            level = nn.Module()
            level.block, level.attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_mult[lvl], ch * ch_mult[lvl]
            for _ in range(num_res_blocks):
                level.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if lvl != self.num_resolutions - 1:
                level.downsample = Downsample(block_in)
            self.down.append(level)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels, 3, padding=1)
        self.quant_conv = nn.Conv2d(2 * z_channels, 2 * z_channels, 1)

    def forward(self, x, wvs=None):
        if self.use_dynamic_ops:
            assert wvs is not None, 'wvs must be provided for Dynamic Encoder'
            h = self.conv_in(x, wvs)
        else:
            h = self.conv_in(x)
        for lvl in range(self.num_resolutions):
            for blk in self.down[lvl].block:
                h = blk(h)
            if lvl != self.num_resolutions - 1:
                h = self.down[lvl].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        h = self.conv_out(swish(self.norm_out(h)))
        return self.quant_conv(h)


class Decoder(nn.Module):  # model.py:223-358
    def __init__(self, ch, out_ch, ch_mult, num_res_blocks, resolution, z_channels,
                 use_dynamic_ops=False, dynamic_conv_kwargs=None):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(z_channels, z_channels, 1)
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.z_channels, self.resolution, self.use_dynamic_ops = z_channels, resolution, use_dynamic_ops
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            level = nn.Module()
            level.block, level.attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[lvl]
            for _ in range(num_res_blocks + 1):
                level.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if lvl != 0:
                level.upsample = Upsample(block_in)
            self.up.insert(0, level)
        self.norm_out = _gn(block_in)
        if use_dynamic_ops:
            wv, inter, kw = _dyn_kwargs(dynamic_conv_kwargs)
            self.conv_out = DynamicConv_decoder(wv_planes=wv, inter_dim=inter, embed_dim=block_in, **kw)
        else:
            self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)

    def forward(self, z, wvs=None):
        h = self.conv_in(self.post_quant_conv(z))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for lvl in reversed(range(self.num_resolutions)):
            for blk in self.up[lvl].block:
                h = blk(h)
            if lvl != 0:
                h = self.up[lvl].upsample(h)
        h = swish(self.norm_out(h))
        if self.use_dynamic_ops:
            assert wvs is not None, 'wvs must be provided for Dynamic Decoder'
            return self.conv_out(h, wvs)
        return self.conv_out(h)


# --------------------------------------------------------------------------
# posterior  (eo_vae/models/modules/distributions.py)
# --------------------------------------------------------------------------


class DiagonalGaussianDistribution:  # distributions.py:19-102
    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, noise=None):
        # reference draws the noise on the CPU default generator (distributions.py:44-46)
        if noise is None:
            noise = torch.randn(self.mean.shape)
        return self.mean + self.std * noise.to(self.mean.device)

    def mode(self):
        return self.mean

    def kl(self):  # distributions.py:64-67
        return 0.5 * torch.sum(self.mean ** 2 + self.var - 1.0 - self.logvar, dim=[1, 2, 3])


# --------------------------------------------------------------------------
# EOFluxVAE  (eo_vae/models/new_autoencoder.py) -- arithmetic only
# --------------------------------------------------------------------------


def pixel_unshuffle2(z):  # 'c (i pi) (j pj) -> (c pi pj) i j', new_autoencoder.py:466-468
    b, c, h, w = z.shape
    return z.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)


def pixel_shuffle2(z):  # '(c pi pj) i j -> c (i pi) (j pj)', new_autoencoder.py:426-428
    b, c4, h, w = z.shape
    c = c4 // 4
    return z.reshape(b, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(b, c, h * 2, w * 2)


class EOFluxVAE(nn.Module):
    def __init__(self, encoder, decoder):  # new_autoencoder.py:76-141
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.bn_eps = 1e-4
        self.bn = nn.BatchNorm2d(4 * encoder.z_channels, affine=False, track_running_stats=True)

    def encode(self, x, wvs):  # :418-421
        return DiagonalGaussianDistribution(self.encoder(x, wvs))

    def _normalize_latent(self, z):  # :533-536 -- batch stats in training, running stats in eval
        self.bn.train() if self.training else self.bn.eval()
        return self.bn(z)

    def _inv_normalize_latent(self, z):  # :538-543 -- ALWAYS running stats, eps 1e-4, forces bn.eval()
        self.bn.eval()
        s = torch.sqrt(self.bn.running_var.view(1, -1, 1, 1) + self.bn_eps)
        return z * s + self.bn.running_mean.view(1, -1, 1, 1)

    def decode(self, z, wvs):  # :423-429
        return self.decoder(pixel_shuffle2(self._inv_normalize_latent(z)), wvs)

    def decode_raw(self, z, wvs):  # :431-438
        return self.decoder(z, wvs)

    def _apply_scale(self, z, scale):  # :520-531 -- new size rounded to a multiple of the 2x2 packing
        h, w = z.shape[-2:]
        sy, sx = scale if isinstance(scale, (tuple, list)) else (scale, scale)
        size = (round(h * sy / 2) * 2, round(w * sx / 2) * 2)
        return F.interpolate(z, size=size, mode='bilinear', align_corners=False)

    def forward(self, x, wvs, sample_posterior=True, scale=None, angle=None, noise=None):  # :447-478
        posterior = self.encode(x, wvs)
        z = posterior.sample(noise) if sample_posterior else posterior.mode()
        if scale is not None:  # EQ-VAE latent transforms (:456-460)
            z = self._apply_scale(z, scale)
        if angle is not None:
            z = torch.rot90(z, k=angle, dims=[-1, -2])
        zn = self._normalize_latent(pixel_unshuffle2(z))
        return self.decode(zn, wvs), posterior

    @staticmethod
    def eqvae_target(images, size, angle=None):  # training_step :612-627 (no_grad in the reference)
        t = F.interpolate(images, size=size, mode='area')
        return t if angle is None else torch.rot90(t, k=angle, dims=[-1, -2])

    @torch.no_grad()
    def reconstruct(self, x, wvs):  # :724-728
        return self.forward(x, wvs, sample_posterior=False)[0]

    @torch.no_grad()
    def encode_to_latent(self, x, wvs):  # :730-738
        return self._normalize_latent(pixel_unshuffle2(self.encode(x, wvs).mode()))

    @torch.no_grad()
    def encode_spatial_normalized(self, x, wvs):  # :480-502
        return pixel_shuffle2(self.encode_to_latent(x, wvs))

    @torch.no_grad()
    def decode_spatial_normalized(self, z, wvs):  # :504-518
        return self.decode(pixel_unshuffle2(z), wvs)


def build_vae(cfg: dict) -> EOFluxVAE:
    """cfg = {'encoder': {...Encoder kwargs}, 'decoder': {...Decoder kwargs}}."""
    return EOFluxVAE(Encoder(**cfg['encoder']), Decoder(**cfg['decoder']))


def model_cfg(ch=128, ch_mult=(1, 2, 4, 4), nrb=2, z=32, res=256, wv_planes=256, num_layers=4,
              channels=3):
    """configs/eo-vae.yaml:33-57 by default; tiny variants for tests via arguments."""
    dyn = {'num_layers': num_layers, 'wv_planes': wv_planes}
    return {
        'encoder': dict(resolution=res, in_channels=channels, ch=ch, ch_mult=list(ch_mult),
                        num_res_blocks=nrb, z_channels=z, use_dynamic_ops=True,
                        dynamic_conv_kwargs=dict(dyn)),
        'decoder': dict(ch=ch, out_ch=channels, ch_mult=list(ch_mult), num_res_blocks=nrb,
                        resolution=res, z_channels=z, use_dynamic_ops=True,
                        dynamic_conv_kwargs=dict(dyn)),
    }


TINY = dict(ch=32, ch_mult=(1, 2, 4, 4), nrb=1, z=32, res=64, wv_planes=64, num_layers=1)

# --------------------------------------------------------------------------
# losses  (eo_vae/models/modules/consistency_loss.py)
# --------------------------------------------------------------------------


def charbonnier(pred, target, eps=1e-3):  # consistency_loss.py:12-21
    d = pred - target
    return torch.mean(torch.sqrt(d * d + eps * eps))


def sam_loss(pred, target, eps=1e-8):  # consistency_loss.py:186-209 -- mean(1 - cos) over pixels
    dot = (pred * target).sum(1)
    den = pred.norm(dim=1) * target.norm(dim=1) + eps
    return (1.0 - dot / den).mean()


def gradient_difference_loss(pred, target, alpha=1.0):  # consistency_loss.py:243-269
    def mag(t, axis):
        a = t[:, :, 1:, :] - t[:, :, :-1, :] if axis == 2 else t[:, :, :, 1:] - t[:, :, :, :-1]
        return a.abs()
    ly = (mag(pred, 2) - mag(target, 2)).abs().pow(alpha)
    lx = (mag(pred, 3) - mag(target, 3)).abs().pow(alpha)
    return lx.mean() + ly.mean()


def _patch_spectrum(x, patch_factor):  # ffl.py:36-57 -- non-overlapping patches, ortho 2-D FFT
    x = x.float()
    b, c, h, w = x.shape
    ph, pw = h // patch_factor, w // patch_factor
    x = x[:, :, :ph * patch_factor, :pw * patch_factor]
    y = x.reshape(b, c, patch_factor, ph, patch_factor, pw).permute(0, 2, 4, 1, 3, 5)
    f = torch.fft.fft2(y.reshape(b, patch_factor * patch_factor, c, ph, pw), norm='ortho')
    return torch.nan_to_num(torch.stack([f.real, f.imag], -1), nan=0.0, posinf=1e6, neginf=-1e6)


def focal_frequency_loss(pred, target, alpha=1.0, patch_factor=2, log_matrix=True,
                         batch_matrix=True):  # ffl.py:59-102 with the flags of consistency_loss.py:388-395
    fp, ft = _patch_spectrum(pred, patch_factor), _patch_spectrum(target, patch_factor)
    d2 = ((fp - ft) ** 2).sum(-1)
    m = torch.sqrt(d2 + 1e-8) ** alpha
    if log_matrix:
        m = torch.log1p(m)
    top = m.max() if batch_matrix else m.flatten(3).max(-1).values[..., None, None]
    top = torch.where(torch.isfinite(top) & (top > 0), top, torch.ones_like(top))
    weight = (m / top).clamp(0.0, 1.0).detach()
    return (weight * d2).mean()


MSSSIM_BETAS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)  # consistency_loss.py:33


def gaussian_taps(sigma=1.5):
    """torchmetrics derives the window from sigma (the kernel_size=5 argument of
    consistency_loss.py:32 is ignored for a Gaussian window): k = 2*int(3.5s+.5)+1 = 11."""
    k = 2 * int(3.5 * sigma + 0.5) + 1
    d = torch.arange((1 - k) / 2.0, (1 + k) / 2.0, 1.0)
    g = torch.exp(-0.5 * (d / sigma) ** 2)
    return g / g.sum()


# torchmetrics behaviour this restatement assumes (SURVEY 8c asks for it to be a named option): releases
# >= 1.0 clamp the two variances at zero before forming cs / ssim (`torch.clamp(..., min=0.0)` in
# functional/image/ssim.py::_ssim_update); earlier releases did not.  The product kernels follow the
# default.  tests/test_oracle_golden.py cross-checks both settings against an independent fp64 / scipy
# derivation; neither is a reference pin (torchmetrics is not installed here, not vendored, not pinned).
MSSSIM_CLAMP_VARIANCE = True


def _ssim_cs(x, y, taps, c1, c2, clamp_variance=None):
    """One scale: per-image mean SSIM and contrast-sensitivity. PARITY UNPINNED
    (restates torchmetrics.functional.image.ssim._ssim_update)."""
    clamp_variance = MSSSIM_CLAMP_VARIANCE if clamp_variance is None else clamp_variance
    b, c, h, w = x.shape
    pad = (taps.numel() - 1) // 2
    xp = F.pad(x, (pad, pad, pad, pad), mode='reflect')
    yp = F.pad(y, (pad, pad, pad, pad), mode='reflect')
    k2d = (taps[:, None] * taps[None, :]).to(x.dtype).expand(c, 1, -1, -1).contiguous()
    stack = torch.cat([xp, yp, xp * xp, yp * yp, xp * yp])
    out = F.conv2d(stack, k2d, groups=c).split(b)
    mx, my = out[0], out[1]
    sxx, syy = out[2] - mx * mx, out[3] - my * my
    if clamp_variance:
        sxx, syy = torch.clamp(sxx, min=0.0), torch.clamp(syy, min=0.0)
    sxy = out[4] - mx * my
    upper = 2 * sxy + c2
    lower = sxx + syy + c2
    ssim = ((2 * mx * my + c1) * upper) / ((mx * mx + my * my + c1) * lower)
    cs = upper / lower
    ssim = ssim[..., pad:-pad, pad:-pad].reshape(b, -1).mean(-1)
    cs = cs[..., pad:-pad, pad:-pad].reshape(b, -1).mean(-1)
    return ssim, cs


def msssim(pred, target, data_range=6.0, betas=MSSSIM_BETAS, sigma=1.5, clamp_variance=None):
    """Batch-mean MS-SSIM, normalize='relu' (consistency_loss.py:24-37 -> torchmetrics).
    PARITY UNPINNED: torchmetrics is not vendored/pinned/installed."""
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    taps = gaussian_taps(sigma)
    x, y = pred, target
    vals = []
    for s in range(len(betas)):
        ssim, cs = _ssim_cs(x, y, taps, c1, c2, clamp_variance)
        vals.append(torch.relu(ssim) if s == len(betas) - 1 else torch.relu(cs))
        x, y = F.avg_pool2d(x, 2), F.avg_pool2d(y, 2)
    stack = torch.stack(vals)  # [S, B]
    b = torch.tensor(betas, dtype=stack.dtype).view(-1, 1)
    return torch.prod(stack ** b, dim=0).mean()


class EOConsistencyLoss(nn.Module):  # consistency_loss.py:329-483 (all branches but the DOFA feature loss)
    def __init__(self, pixel_weight=1.0, rec_loss_type='l1', spectral_weight=0.0, spatial_weight=0.0,
                 freq_weight=0.0, msssim_weight=0.0, spectral_start_step=0, spatial_start_step=0,
                 freq_start_step=0, msssim_start_step=0, patch_factor=2, ffl_alpha=1.0,
                 kl_weight=0.0, **unused):
        super().__init__()
        self.pixel_weight, self.rec_loss_type = pixel_weight, rec_loss_type
        self.msssim_weight, self.msssim_start_step = msssim_weight, msssim_start_step
        self.spectral_weight, self.spectral_start_step = spectral_weight, spectral_start_step
        self.spatial_weight, self.spatial_start_step = spatial_weight, spatial_start_step
        self.freq_weight, self.freq_start_step = freq_weight, freq_start_step
        self.patch_factor, self.ffl_alpha = patch_factor, ffl_alpha
        self.kl_weight = kl_weight  # opt-in extension (absent in reference)

    def forward(self, inputs, wvs, reconstructions, global_step=0, split='train',
                posterior=None, **kw):
        logs = {}
        total = torch.tensor(0.0, device=inputs.device)
        if self.pixel_weight > 0:
            if self.rec_loss_type == 'l1':
                l_rec = F.l1_loss(reconstructions, inputs)
            elif self.rec_loss_type == 'char':
                l_rec = charbonnier(reconstructions, inputs)
            else:
                raise ValueError("rec_loss_type must be 'l1' or 'char'")
            total = total + self.pixel_weight * l_rec
            logs[f'{split}/loss_rec'] = l_rec.detach()
        if self.spectral_weight > 0 and global_step >= self.spectral_start_step:  # :423-428
            l_sam = sam_loss(reconstructions, inputs)
            total = total + self.spectral_weight * l_sam
            logs[f'{split}/loss_spectral'] = l_sam.detach()
        if self.spatial_weight > 0 and global_step >= self.spatial_start_step:  # :430-435
            l_sp = gradient_difference_loss(reconstructions, inputs)
            total = total + self.spatial_weight * l_sp
            logs[f'{split}/loss_spatial'] = l_sp.detach()
        if self.freq_weight > 0 and global_step >= self.freq_start_step:  # :437-464, 1000-step linear warm-up
            raw = focal_frequency_loss(reconstructions, inputs, self.ffl_alpha, self.patch_factor)
            w_now = self.freq_weight * min(1.0, max(0.0, (global_step - self.freq_start_step) / 1000))
            total = total + raw * w_now
            logs[f'{split}/loss_freq_raw'] = raw.detach()
            logs[f'{split}/ffl_weight'] = torch.tensor(w_now)
        if self.msssim_weight > 0 and global_step >= self.msssim_start_step:
            l_ms = 1.0 - msssim(reconstructions, inputs)
            total = total + self.msssim_weight * l_ms
            logs[f'{split}/loss_msssim'] = l_ms.detach()
        if self.kl_weight > 0 and posterior is not None:
            l_kl = posterior.kl().mean()
            total = total + self.kl_weight * l_kl
            logs[f'{split}/loss_kl'] = l_kl.detach()
        logs[f'{split}/loss_total'] = total.detach()
        return total, logs


# --------------------------------------------------------------------------
# optimiser step  (new_autoencoder.py:36-56, 549-585, 648-655)
# --------------------------------------------------------------------------


def lr_factor(step, warmup_steps, total_steps, base_lr, final_lr):  # :36-56
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    cosine = 0.5 * (1.0 + math.cos(math.pi * progress))
    return ((base_lr - final_lr) * cosine + final_lr) / base_lr


class TrainState:
    """Adam(lr=base_lr) + clip_grad_norm_ + per-step LambdaLR, as training_step does."""

    def __init__(self, model, base_lr=1e-4, final_lr=None, warmup_epochs=None,
                 decay_end_epoch=None, clip_grad=None, steps_per_epoch=2000):
        self.model = model
        self.params = [p for p in model.encoder.parameters() if p.requires_grad] + \
                      [p for p in model.decoder.parameters() if p.requires_grad]
        self.opt = torch.optim.Adam(self.params, lr=base_lr)
        self.clip_grad = clip_grad
        self.sched = None
        if all([final_lr, warmup_epochs, decay_end_epoch]):
            wu, tot = warmup_epochs * steps_per_epoch, decay_end_epoch * steps_per_epoch
            self.sched = torch.optim.lr_scheduler.LambdaLR(
                self.opt, lambda s: lr_factor(s, wu, tot, base_lr, final_lr))
        self.global_step = 0

    def step(self, loss_fn, images, wvs, noise=None, scale=None, angle=None):
        """scale/angle: the EQ-VAE draw of training_step (:601-627), injected instead of random."""
        recon, post = self.model(images, wvs, noise=noise, scale=scale, angle=angle)
        if scale is not None:
            with torch.no_grad():
                images = self.model.eqvae_target(images, recon.shape[-2:], angle)
        self.opt.zero_grad()
        loss, logs = loss_fn(inputs=images, wvs=wvs, reconstructions=recon,
                             global_step=self.global_step, split='train', posterior=post)
        loss.backward()
        gnorm = None
        if self.clip_grad:
            gnorm = torch.nn.utils.clip_grad_norm_(self.params, self.clip_grad)
        logs['train/lr'] = self.opt.param_groups[0]['lr']
        self.opt.step()
        if self.sched:
            self.sched.step()
        self.global_step += 1
        return loss.detach(), logs, gnorm


# --------------------------------------------------------------------------
# Stage-1 weight distillation  (weight_distill_train.py:146-331)
# --------------------------------------------------------------------------


class DistillState:
    """Data-free fit of the two dynamic layers to a teacher's RGB conv_in / conv_out kernels:
    MSE(weight) + MSE(bias), AdamW(lr, weight_decay 1e-5) over conv_in + conv_out parameters,
    CosineAnnealingLR(T_max = max_steps, eta_min = lr / 100) stepped every iteration."""

    def __init__(self, encoder, decoder, teacher, rgb_wavelengths=(0.665, 0.560, 0.490), lr=1e-4,
                 max_steps=5000, weight_loss_scale=1.0, bias_loss_scale=1.0):
        self.encoder, self.decoder, self.teacher = encoder, decoder, teacher
        self.wvs = torch.tensor(rgb_wavelengths, dtype=torch.float32)
        self.w_scale, self.b_scale = weight_loss_scale, bias_loss_scale
        params = list(encoder.conv_in.parameters()) + list(decoder.conv_out.parameters())  # :306-312
        self.opt = torch.optim.AdamW(params, lr=lr, weight_decay=1e-5)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=max_steps, eta_min=lr * 0.01)

    def loss(self):  # compute_distillation_loss, :190-264
        logs, total = {}, torch.tensor(0.0)
        for tag, layer in (('enc', self.encoder.conv_in), ('dec', self.decoder.conv_out)):
            w, b = layer.get_distillation_weight(self.wvs)
            tw, tb = self.teacher[f'{tag}oder_weight'], self.teacher[f'{tag}oder_bias']
            lw = F.mse_loss(w, tw)
            total = total + lw * self.w_scale
            logs[f'{tag}_weight_loss'] = lw
            if b is not None and tb is not None:
                lb = F.mse_loss(b, tb)
                total = total + lb * self.b_scale
                logs[f'{tag}_bias_loss'] = lb
            with torch.no_grad():
                logs[f'{tag}_weight_mae'] = F.l1_loss(w, tw)
                logs[f'{tag}_weight_max_err'] = (w - tw).abs().max()
        logs['total_loss'] = total
        return total, logs

    def step(self):
        loss, logs = self.loss()
        self.opt.zero_grad()
        loss.backward()
        lr = self.opt.param_groups[0]['lr']
        self.opt.step()
        self.sched.step()
        return loss.detach(), {k: v.detach() for k, v in logs.items()}, lr


# --------------------------------------------------------------------------
# encode_latents running statistics  (encode_latents.py:43-113)
# --------------------------------------------------------------------------


class RunningStats:
    def __init__(self, channels, dims=(0, 2, 3)):
        self.mean = torch.zeros(channels)
        self.var = torch.ones(channels)
        self.std = torch.ones(channels)
        self.count = torch.zeros(1)
        self.min = torch.full((channels,), float('inf'))
        self.max = torch.full((channels,), float('-inf'))
        self.dims = dims

    @torch.no_grad()
    def update(self, x):  # encode_latents.py:65-94 (Chan merge; torch.var is unbiased)
        bm, bv = torch.mean(x, dim=self.dims), torch.var(x, dim=self.dims)
        bc = 1.0
        for d in self.dims:
            bc *= x.shape[d]
        bc = torch.tensor(bc, dtype=torch.float)
        n = self.count + bc
        delta = bm - self.mean
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / (n + 1e-8)
        self.mean = (self.mean * self.count + bm * bc) / n
        self.var = m2 / n
        self.count = n
        self.std = torch.sqrt(self.var + 1e-8)
        self.min = torch.minimum(self.min, torch.amin(x, dim=self.dims))
        self.max = torch.maximum(self.max, torch.amax(x, dim=self.dims))
