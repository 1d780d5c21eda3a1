"""EOFluxVAE on the MI355X hot path.

Drop-in for ``eo_vae.models.new_autoencoder.EOFluxVAE`` (reference file
eo_vae/models/new_autoencoder.py): same constructor / from_config / from_pretrained signatures,
same state_dict keys, same training_step contract (manual optimisation).  The arithmetic is the
HIP library's; this file is host glue."""
from __future__ import annotations

import math
import os
import random
from typing import Any

import torch
from torch.optim.lr_scheduler import LambdaLR

from . import ops
from .distributions import DiagonalGaussianDistribution
from .lightning_compat import LightningModule
from .model import Decoder, Encoder


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int,
                                    base_lr: float, final_lr: float, num_cycles: float = 0.5):
    """Linear warm-up from factor 0, then cosine from base_lr to final_lr (reference :36-56)."""

    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return step / max(1, num_warmup_steps)
        t = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
        cos = 0.5 * (1.0 + math.cos(2.0 * math.pi * num_cycles * t))
        return (final_lr + (base_lr - final_lr) * cos) / base_lr

    return LambdaLR(optimizer, factor)


_VAE_KWARGS = ('freeze_body', 'base_lr', 'final_lr', 'warmup_epochs', 'decay_end_epoch', 'clip_grad',
               'p_prior', 'p_prior_s', 'anisotropic', 'latent_noise_p', 'noise_tau', 'image_key')


def _pack2x2(z):  # '... c (i pi) (j pj) -> ... (c pi pj) i j'
    *lead, c, h, w = z.shape
    z = z.reshape(*lead, c, h // 2, 2, w // 2, 2)
    n = len(lead)
    return z.permute(*range(n), n, n + 2, n + 4, n + 1, n + 3).reshape(*lead, c * 4, h // 2, w // 2)


def _unpack2x2(z):  # '... (c pi pj) i j -> ... c (i pi) (j pj)'
    *lead, c4, h, w = z.shape
    z = z.reshape(*lead, c4 // 4, 2, 2, h, w)
    n = len(lead)
    return z.permute(*range(n), n, n + 3, n + 1, n + 4, n + 2).reshape(*lead, c4 // 4, h * 2, w * 2)


class EOFluxVAE(LightningModule):
    def __init__(self, encoder: torch.nn.Module, decoder: torch.nn.Module, loss_fn: torch.nn.Module,
                 ckpt_path: str | None = None, ignore_keys: list[str] | None = None,
                 freeze_body: bool = True, base_lr: float = 1e-4, final_lr: float | None = None,
                 warmup_epochs: int | None = None, decay_end_epoch: int | None = None,
                 clip_grad: float | None = None, p_prior: float = 0.0, p_prior_s: float = 0.0,
                 anisotropic: bool = False, latent_noise_p: float = 0.0, noise_tau: float = 0.8,
                 image_key: str = 'image') -> None:
        super().__init__()
        self.encoder, self.decoder, self.loss_fn = encoder, decoder, loss_fn
        self.image_key = image_key
        self.base_lr, self.final_lr = base_lr, final_lr
        self.warmup_epochs, self.decay_end_epoch, self.clip_grad = warmup_epochs, decay_end_epoch, clip_grad
        self.p_prior, self.p_prior_s, self.anisotropic = p_prior, p_prior_s, anisotropic
        self.latent_noise_p, self.noise_tau = latent_noise_p, noise_tau
        self.ps = [2, 2]
        self.bn_eps = 1e-4  # inverse normalisation only; the BatchNorm itself keeps eps 1e-5 (quirk q3)
        self.bn = torch.nn.BatchNorm2d(math.prod(self.ps) * encoder.z_channels, affine=False,
                                       track_running_stats=True)
        self.automatic_optimization = False
        self.freeze_body = freeze_body
        if freeze_body:
            self._freeze_body()
        if ckpt_path:
            self._load_checkpoint(ckpt_path, ignore_keys or [])

    # ------------------------------------------------------------------ construction
    @staticmethod
    def _read_config_file(config_path: str) -> dict[str, Any]:
        if not os.path.exists(config_path):
            raise FileNotFoundError(f'Config file not found: {config_path}')
        try:
            from omegaconf import OmegaConf  # used when present, like the reference (:143-153)
            data = OmegaConf.to_container(OmegaConf.load(config_path), resolve=True)
        except ImportError:
            from .config import load_config
            data = load_config(config_path)
        if not isinstance(data, dict):
            raise ValueError('Model config must deserialize to a dictionary')
        return data

    @staticmethod
    def _extract_model_sections(config: dict[str, Any]):
        """Accepts a full training YAML (`model:` key) or a bare {encoder, decoder} mapping."""
        model_cfg = config.get('model', config)
        if not isinstance(model_cfg, dict):
            raise ValueError('Invalid config: `model` section must be a dictionary')
        if 'encoder' not in model_cfg or 'decoder' not in model_cfg:
            raise ValueError('Invalid config: expected `encoder` and `decoder` sections')
        enc = {k: v for k, v in model_cfg['encoder'].items() if k != '_target_'}
        dec = {k: v for k, v in model_cfg['decoder'].items() if k != '_target_'}
        vae = {k: model_cfg[k] for k in _VAE_KWARGS if k in model_cfg}
        return enc, dec, vae

    @classmethod
    def from_config(cls, config_path: str, ckpt_path: str | None = None, *,
                    loss_fn: torch.nn.Module | None = None, freeze_body: bool | None = None,
                    ignore_keys: list[str] | None = None, device=None, eval_mode: bool = True):
        enc_cfg, dec_cfg, vae_kwargs = cls._extract_model_sections(cls._read_config_file(config_path))
        if freeze_body is not None:
            vae_kwargs['freeze_body'] = freeze_body
        model = cls(encoder=Encoder(**enc_cfg), decoder=Decoder(**dec_cfg),
                    loss_fn=loss_fn if loss_fn is not None else torch.nn.Identity(),
                    freeze_body=vae_kwargs.pop('freeze_body', False), **vae_kwargs)
        if ckpt_path:
            model._load_checkpoint(ckpt_path, ignore_keys or [])
        if device is not None:
            model = model.to(device)
        if eval_mode:
            model.eval()
        return model

    @classmethod
    def from_pretrained(cls, repo_id: str, *, ckpt_filename: str = 'eo-vae.ckpt',
                        config_filename: str = 'model_config.yaml', revision: str | None = None,
                        cache_dir: str | None = None, local_files_only: bool = False,
                        loss_fn: torch.nn.Module | None = None, freeze_body: bool | None = None,
                        ignore_keys: list[str] | None = None, device=None, eval_mode: bool = True):
        try:
            from huggingface_hub import hf_hub_download
        except ImportError as exc:
            raise ImportError('huggingface_hub is required for from_pretrained') from exc
        fetch = dict(repo_id=repo_id, revision=revision, cache_dir=cache_dir,
                     local_files_only=local_files_only)
        config_path = hf_hub_download(filename=config_filename, **fetch)
        ckpt_path = hf_hub_download(filename=ckpt_filename, **fetch)
        return cls.from_config(config_path=config_path, ckpt_path=ckpt_path, loss_fn=loss_fn,
                               freeze_body=freeze_body, ignore_keys=ignore_keys, device=device,
                               eval_mode=eval_mode)

    def _freeze_body(self) -> None:
        print('Freezing VAE body (dynamic layers remain trainable)')
        for net in (self.encoder, self.decoder):
            for p in net.parameters():
                p.requires_grad = False
        if self.encoder.use_dynamic_ops:
            for p in self.encoder.conv_in.parameters():
                p.requires_grad = True
            print('  -> Encoder conv_in: TRAINABLE')
        if self.decoder.use_dynamic_ops:
            for p in self.decoder.conv_out.parameters():
                p.requires_grad = True
            print('  -> Decoder conv_out: TRAINABLE')

    # ------------------------------------------------------------------ checkpoints
    def _load_checkpoint(self, path: str, ignore_keys: list[str]) -> None:
        """Three formats (:295-356): distilled .pt (dynamic layers only), Flux .safetensors (body),
        full Lightning .ckpt / plain state_dict."""
        if not os.path.exists(path):
            print(f'Checkpoint not found: {path}')
            return
        print(f'Loading weights from {path}')
        if path.endswith('.pt'):
            ckpt = torch.load(path, map_location='cpu')
            if 'encoder_conv_in_state_dict' in ckpt or 'decoder_conv_out_state_dict' in ckpt:
                self._load_distilled_checkpoint(ckpt)
                return
        if path.endswith('.safetensors'):
            from safetensors import safe_open
            with safe_open(path, framework='pt', device='cpu') as f:
                sd = {k: f.get_tensor(k) for k in f.keys()}
        else:
            sd = torch.load(path, map_location='cpu')
            sd = sd.get('state_dict', sd)

        def is_static_io(k: str) -> bool:  # static conv_in/conv_out tensors of a non-dynamic VAE
            dyn = 'weight_generator' in k or 'fclayer' in k
            return ((self.encoder.use_dynamic_ops and 'encoder.conv_in' in k and not dyn)
                    or (self.decoder.use_dynamic_ops and 'decoder.conv_out' in k and not dyn))

        sd = {k: v for k, v in sd.items()
              if not is_static_io(k) and not any(k.startswith(ik) for ik in ignore_keys)}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        self._verify_loading(missing, unexpected, ignore_keys)

    def _load_distilled_checkpoint(self, ckpt: dict) -> None:
        print('Detected distillation checkpoint format')
        if self.encoder.use_dynamic_ops and ckpt.get('encoder_conv_in_state_dict'):
            self.encoder.conv_in.load_state_dict(ckpt['encoder_conv_in_state_dict'])
            print('  Loaded encoder.conv_in from distilled checkpoint')
        if self.decoder.use_dynamic_ops and ckpt.get('decoder_conv_out_state_dict'):
            self.decoder.conv_out.load_state_dict(ckpt['decoder_conv_out_state_dict'])
            print('  Loaded decoder.conv_out from distilled checkpoint')
        if 'distill_config' in ckpt:
            print(f'  Distillation loss was: {ckpt["distill_config"].get("final_loss", "N/A")}')

    def _verify_loading(self, missing_keys, unexpected_keys, ignore_keys) -> None:
        allowed = list(ignore_keys)
        if self.encoder.use_dynamic_ops:
            allowed.append('encoder.conv_in')
        if self.decoder.use_dynamic_ops:
            allowed.append('decoder.conv_out')
        critical = [k for k in missing_keys if not any(k.startswith(p) for p in allowed)]
        if critical:
            raise RuntimeError(f'Critical weights missing from checkpoint:\n{critical[:20]}...\n'
                               f'Total: {len(critical)} missing keys')
        print(f'Checkpoint loaded: {len(missing_keys)} missing (expected), '
              f'{len(unexpected_keys)} unexpected (ignored)')

    # ------------------------------------------------------------------ forward path
    def encode(self, x, wvs) -> DiagonalGaussianDistribution:
        return DiagonalGaussianDistribution(self.encoder(x, wvs))

    def _normalize_latent(self, z_packed):
        """BatchNorm on the packed latent: batch statistics (and running-stat update) in training,
        running statistics in eval (:533-536)."""
        self.bn.train() if self.training else self.bn.eval()
        z_spatial = _unpack2x2(z_packed)
        return self._normalize_spatial(z_spatial)

    def _normalize_spatial(self, z):
        """Fused pixel-unshuffle + BatchNorm kernel: spatial [B,Z,H,W] -> packed normalised."""
        training = bool(self.training)
        self.bn.train() if training else self.bn.eval()
        zp = ops.LatentNormFn.apply(z, self.bn.running_mean, self.bn.running_var, training,
                                    float(self.bn.eps), float(self.bn.momentum))
        if training:
            self.bn.num_batches_tracked += 1
        return zp

    def _inv_normalize_latent(self, z_packed):
        """Always the running statistics, eps 1e-4, and it leaves the BatchNorm in eval mode
        (:538-543).  Returns the *packed* tensor like the reference."""
        return _pack2x2(self._inv_normalize_to_spatial(z_packed))

    def _inv_normalize_to_spatial(self, z_packed):
        self.bn.eval()
        return ops.LatentInvNormFn.apply(z_packed, self.bn.running_mean, self.bn.running_var,
                                         self.bn_eps)

    def decode(self, z, wvs):
        """z: packed normalised latent [B, 4Z, h, w] (or with extra leading dims)."""
        if z.dim() != 4:
            lead = z.shape[:-3]
            out = self.decode(z.reshape(-1, *z.shape[-3:]), wvs)
            return out.reshape(*lead, *out.shape[1:])
        return self.decoder(self._inv_normalize_to_spatial(z), wvs)

    def decode_raw(self, z, wvs):
        return self.decoder(z, wvs)

    # Posterior noise, prefetched.  The reference draws it from the CPU default generator inside sample()
    # (distributions.py:44-46) and copies it from pageable memory: a blocking copy that makes the host wait for
    # the encoder and leaves the GPU idle (~3 ms per step) until the host has caught up again.  Here the SAME
    # draw (same generator, same number stream: nothing else consumes it in between) is made before the
    # encoder is enqueued, into one of two pinned buffers, and copied without blocking.
    _lat_geom = None          # (z_channels, downsample factor y, x), learnt on the first sampled forward
    _noise_slots = None
    _noise_ready = None       # event of the copy stream behind the latest prefetched noise

    def _prefetch_noise(self, x):
        if self._lat_geom is None or not x.is_cuda:
            return None
        z, fy, fx = self._lat_geom
        shape = (x.shape[0], z, x.shape[2] // fy, x.shape[3] // fx)
        if self._noise_slots is None or self._noise_slots['shape'] != shape:
            self._noise_slots = {'shape': shape, 'i': 0, 'ev': [None, None],
                                 'buf': [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2)]}
        st = self._noise_slots
        i = st['i'] = st['i'] ^ 1
        if st['ev'][i] is not None:
            st['ev'][i].synchronize()       # the copy that last read this buffer has run
        torch.randn(shape, out=st['buf'][i])
        # the copy rides on its own stream (50 us of PCIe time that would otherwise sit in front of the encoder on the compute
        # stream); forward() makes the compute stream wait for it just before the posterior is sampled
        if st.get('stream') is None:
            st['stream'] = torch.cuda.Stream(device=x.device)
            from .parallel import SIDE_STREAMS
            SIDE_STREAMS.append(st['stream'])
        with torch.cuda.stream(st['stream']):
            dev = st['buf'][i].to(x.device, non_blocking=True)
            st['ev'][i] = torch.cuda.Event()
            st['ev'][i].record()
        self._noise_ready = st['ev'][i]
        return dev

    def noising(self, x):
        sigma = self.noise_tau * torch.rand((x.size(0),) + (1,) * (x.dim() - 1), device=x.device)
        return x + sigma * torch.randn_like(x)

    def forward(self, x, wvs, sample_posterior: bool = True, scale=None, angle: int | None = None,
                noise: torch.Tensor | None = None):
        """encode -> sample/mode -> (EQ-VAE transform) -> shuffle+BN -> inverse BN+unshuffle -> decode.
        `noise` (optional, beyond the reference signature) injects the posterior noise."""
        prefetched = False
        if sample_posterior and noise is None:
            noise = self._prefetch_noise(x)
            prefetched = noise is not None
        posterior = self.encode(x, wvs)
        if sample_posterior and self._lat_geom is None:
            m = posterior.parameters
            self._lat_geom = (m.shape[1] // 2, x.shape[2] // m.shape[2], x.shape[3] // m.shape[3])
        if prefetched and tuple(noise.shape) != tuple(posterior.mean.shape):
            # the latent geometry is cached from the first sampled forward as floor ratios; an input whose size is not
            # a multiple of the down-sampling factor would get noise of another shape.  Never re-drawn silently: the
            # draw has already advanced the CPU generator the reference's stream depends on.
            raise ops.L.EovaeError(f'posterior noise was prefetched as {tuple(noise.shape)} but the latent is '
                                   f'{tuple(posterior.mean.shape)}: input {tuple(x.shape)} does not fit the cached latent '
                                   'geometry; pass noise= explicitly or set model._lat_geom = None')
        if prefetched:
            main = torch.cuda.current_stream()
            main.wait_event(self._noise_ready)      # the copy stream's H2D of this step's noise
            noise.record_stream(main)
        z = posterior.sample(noise) if sample_posterior else posterior.mode()
        if scale is not None or angle is not None:  # EQ-VAE: bilinear scale + rot90 in ONE gather kernel
            size = self._scaled_size(z, scale) if scale is not None else z.shape[-2:]
            z = ops.resize_rot(z, size, angle)
        z_normalized = self._normalize_spatial(z)
        if self.training and random.random() < self.latent_noise_p:
            z_normalized = self.noising(z_normalized)
        return self.decode(z_normalized, wvs), posterior

    @torch.no_grad()
    def reconstruct(self, x, wvs):
        return self.forward(x, wvs, sample_posterior=False)[0]

    @torch.no_grad()
    def encode_to_latent(self, x, wvs):
        """-> packed normalised latent [B, 4Z, H/16, W/16]."""
        return self._normalize_spatial(self.encode(x, wvs).mode())

    @torch.no_grad()
    def encode_spatial_normalized(self, x, wvs):
        """-> [B, Z, H/8, W/8]: spatial layout with the VAE's latent normalisation applied."""
        return _unpack2x2(self.encode_to_latent(x, wvs))

    @torch.no_grad()
    def decode_spatial_normalized(self, z, wvs):
        return self.decode(_pack2x2(z), wvs)

    def _scaled_size(self, z, scale):
        """(:520-531) target size rounded to a multiple of the 2x2 latent packing."""
        h, w = z.shape[-2:]
        sy, sx = scale if isinstance(scale, (tuple, list)) else (scale, scale)
        return round(h * sy / self.ps[0]) * self.ps[0], round(w * sx / self.ps[1]) * self.ps[1]

    def _apply_scale(self, z, scale):
        return ops.resize_rot(z, self._scaled_size(z, scale), None)

    # ------------------------------------------------------------------ training
    def configure_optimizers(self):
        params = [p for net in (self.encoder, self.decoder) for p in net.parameters() if p.requires_grad]
        from .optim import FusedAdam   # raises EovaeError for CPU parameters: no fallback optimiser
        opt_ae = FusedAdam(params, lr=self.base_lr)
        optimizers, schedulers = [opt_ae], []
        if hasattr(self.loss_fn, 'discriminator'):
            optimizers.append(torch.optim.Adam(self.loss_fn.discriminator.parameters(), lr=self.base_lr))
        if all([self.final_lr, self.warmup_epochs, self.decay_end_epoch]):
            steps_per_epoch = 2000  # the reference's hard-coded estimate (:569)
            for opt in optimizers:
                sch = get_cosine_schedule_with_warmup(
                    opt, num_warmup_steps=self.warmup_epochs * steps_per_epoch,
                    num_training_steps=self.decay_end_epoch * steps_per_epoch,
                    base_lr=self.base_lr, final_lr=self.final_lr)
                schedulers.append({'scheduler': sch, 'interval': 'step'})
        return (optimizers, schedulers) if schedulers else optimizers

    def _clip(self, optimizer):
        if hasattr(optimizer, 'clip_grad_norm'):
            return optimizer.clip_grad_norm(self.clip_grad)
        return torch.nn.utils.clip_grad_norm_(optimizer.param_groups[0]['params'], self.clip_grad)

    def _prepack_weights(self):
        """f16c8 mode, training: pack the plain 3x3 weights of encoder and decoder once, up front (two launches instead
        of ~180).  Rebuilt whenever a weight's storage has moved (optimiser flattening, .to(), a loaded checkpoint)."""
        if not (ops.PREPACK and ops.CONV_PRECISION == 'f16c8' and torch.is_grad_enabled()):
            return
        pp = getattr(self, '_weight_prepack', None)
        if pp is None or pp.stale():
            from .layers import Conv2d
            ws = [m.weight for m in list(self.encoder.modules()) + list(self.decoder.modules())
                  if isinstance(m, Conv2d) and m.mode == 'normal' and m.weight.requires_grad]
            pp = ops.WeightPrepack(ws) if ws and ws[0].is_cuda else None
            self._weight_prepack = pp
        if pp is not None:
            pp.run()

    def training_step(self, batch, batch_idx, noise: torch.Tensor | None = None):
        opts = self.optimizers()
        opt_gen = opts[0] if isinstance(opts, (list, tuple)) else opts
        schs = self.lr_schedulers()
        sch_gen = schs[0] if isinstance(schs, (list, tuple)) and schs else schs

        images, wvs = batch[self.image_key], batch['wvs']
        target = images
        if images.is_cuda:
            ops.raise_on_gn_backward_timeouts()   # a time-out of an earlier step's one-pass GroupNorm backward
        self._prepack_weights()     # every plain 3x3 layer's f16c8 packs from two launches (ops.WeightPrepack)
        try:
            scale_bins = [0.375, 0.5, 0.75]
            if random.random() < self.p_prior:  # EQ-VAE latent equivariance (:605-619)
                angle = random.choice([1, 2, 3])
                scale = ((random.choice(scale_bins), random.choice(scale_bins)) if self.anisotropic
                         else random.choice(scale_bins))
                recon, posterior = self.forward(images, wvs, scale=scale, angle=angle, noise=noise)
                target = ops.area_rot(images, recon.shape[-2:], angle)  # area resize + rot90, one kernel
            elif random.random() < self.p_prior_s:  # prior preservation (:621-627)
                recon, posterior = self.forward(images, wvs, scale=random.choice(scale_bins), noise=noise)
                target = ops.area_rot(images, recon.shape[-2:], None)
            else:
                recon, posterior = self.forward(images, wvs, noise=noise)

            opt_gen.zero_grad()
            loss_kwargs = dict(inputs=target, wvs=wvs, reconstructions=recon, optimizer_idx=0,
                               global_step=self.global_step, last_layer=self.get_last_layer(), split='train')
            if getattr(self.loss_fn, 'weights', {}).get('kl', 0) > 0:
                loss_kwargs['posterior'] = posterior
            gen_loss, logs = self.loss_fn(**loss_kwargs)
            self.manual_backward(gen_loss)
        finally:
            ops.WeightPrepack.release()   # the packs belong to THIS step's weights, whatever happens in it
        if self.clip_grad:
            logs['train/grad_norm'] = self._clip(opt_gen)
        opt_gen.step()
        if sch_gen:
            sch_gen.step()
        logs['train/lr'] = opt_gen.param_groups[0]['lr']
        self.log_dict(logs, prog_bar=True, logger=True, on_step=True, on_epoch=False)
        return gen_loss

    def validation_step(self, batch, batch_idx):
        images, wvs = batch[self.image_key], batch['wvs']
        recon, _ = self.forward(images, wvs)
        val_loss, logs = self.loss_fn(inputs=images, wvs=wvs, reconstructions=recon, optimizer_idx=0,
                                      global_step=self.global_step, last_layer=None, split='val')
        self.log_dict(logs, prog_bar=True, logger=True, on_step=False, on_epoch=True)
        return val_loss

    def get_last_layer(self):
        if hasattr(self.decoder, 'output_conv_weight'):
            return self.decoder.output_conv_weight
        return getattr(self.decoder.conv_out, 'weight', None)
