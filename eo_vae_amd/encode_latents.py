"""Frozen-VAE dataset encoding (reference: encode_latents.py:43-169, 305-352) on the HIP path.

encode_raw / encode_spatial_norm keep the reference's meaning; RunningStats keeps the
RunningStatsButFast state on the GPU (one fused kernel per batch) and adds the cross-rank Chan
merge that a sharded (one process per GPU) run needs."""
from __future__ import annotations

import json
import os

import numpy as np
import torch
import torch.distributed as dist

from . import ops


@torch.no_grad()
def encode_raw(model, img, wvs):
    """Posterior mean straight from the encoder (encode_latents.py:139-157)."""
    return torch.chunk(model.encoder(img, wvs), 2, dim=1)[0]


@torch.no_grad()
def encode_spatial_norm(model, img, wvs):
    """VAE-normalised spatial latent [B, 32, H/8, W/8] (encode_latents.py:160-169)."""
    return model.encode_spatial_normalized(img, wvs)


class RunningStats:
    """Per-channel running mean / var / std / min / max / count over (B, H, W)."""

    def __init__(self, channels: int, device):
        self.channels = channels
        self.state = torch.zeros(5, channels, dtype=torch.float32, device=device)
        self.state[1].fill_(1.0)
        self.state[2].fill_(float('inf'))
        self.state[3].fill_(float('-inf'))

    def update(self, z: torch.Tensor) -> None:
        ops.running_stats_update(z, self.state)

    @staticmethod
    def _merge(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """Chan merge of two (mean, var, min, max, count) states with the same formula the
        per-batch update uses (encode_latents.py:83-93)."""
        ca, cb = a[4], b[4]
        n = ca + cb
        safe = torch.where(n > 0, n, torch.ones_like(n))
        delta = b[0] - a[0]
        mean = (a[0] * ca + b[0] * cb) / safe
        var = (a[1] * ca + b[1] * cb + delta ** 2 * ca * cb / (n + 1e-8)) / safe
        out = torch.stack([mean, var, torch.minimum(a[2], b[2]), torch.maximum(a[3], b[3]), n])
        return torch.where((cb > 0).expand_as(out), out, a)

    def merge_across_ranks(self, group=None) -> None:
        """One all_gather of the 5 x C state + ordered Chan merges (not a plain SUM)."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        world = dist.get_world_size(group)
        parts = [torch.empty_like(self.state) for _ in range(world)]
        dist.all_gather(parts, self.state, group=group)
        merged = parts[0]
        for p in parts[1:]:
            merged = self._merge(merged, p)
        self.state.copy_(merged)

    def get_stats_dict(self) -> dict:
        s = self.state.detach().cpu()
        return {'mean': s[0], 'var': s[1], 'std': torch.sqrt(s[1] + 1e-8), 'min': s[2], 'max': s[3],
                'count': s[4, :1].clone()}


@torch.no_grad()
def encode_split(model, loader, encode_fn=encode_spatial_norm, out_dir=None, rank=0, key='image'):
    """Encode every batch of `loader`; returns the RunningStats.  Latents are written as one
    .npz per batch when out_dir is given (device->host copy overlaps the next batch's encode)."""
    device = next(model.parameters()).device
    stats = None
    pending = None
    for i, batch in enumerate(loader):
        img, wvs = batch[key].to(device), batch['wvs'].to(device)
        z = encode_fn(model, img, wvs)
        if stats is None:
            stats = RunningStats(z.shape[1], device)
        stats.update(z)
        if out_dir is not None:
            host = torch.empty(z.shape, dtype=z.dtype, pin_memory=True)
            host.copy_(z, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            if pending is not None:
                _flush(*pending)
            pending = (host, ev, os.path.join(out_dir, f'latents_r{rank}_{i:06d}.npz'))
    if pending is not None:
        _flush(*pending)
    return stats


def _flush(host, ev, path):
    ev.synchronize()
    np.savez_compressed(path, latent=host.numpy())


@torch.no_grad()
def encode_split_pairs(model, loader, output_dir, wvs_lr, wvs_hr, stats_lr=None, stats_hr=None,
                       encode_fn=encode_raw):
    """The reference's paired (low-res / high-res sensor) contract (encode_latents.py:305-352): batches
    carry 'image_lr', 'image_hr', 'aoi'; one `<aoi>.npz` per sample with keys lr_latent, hr_latent,
    lr_image, hr_image -- the layout its Sen2NaipCrossSensorLatent dataset reads back.  Device->host
    copies of batch i overlap the encode of batch i+1 (pinned staging + event)."""
    device = next(model.parameters()).device
    os.makedirs(output_dir, exist_ok=True)
    wvs_lr, wvs_hr = wvs_lr.to(device), wvs_hr.to(device)
    pending = None

    def write(aois, tensors, ev):
        ev.synchronize()
        z_lr, z_hr, x_lr, x_hr = (t.numpy() for t in tensors)
        for i, aoi in enumerate(aois):
            np.savez_compressed(os.path.join(output_dir, f'{aoi}.npz'), lr_latent=z_lr[i], hr_latent=z_hr[i],
                                lr_image=x_lr[i], hr_image=x_hr[i])

    for batch in loader:
        lr, hr = batch['image_lr'].to(device), batch['image_hr'].to(device)
        z_lr, z_hr = encode_fn(model, lr, wvs_lr), encode_fn(model, hr, wvs_hr)
        if stats_lr is None:
            stats_lr, stats_hr = RunningStats(z_lr.shape[1], device), RunningStats(z_hr.shape[1], device)
        stats_lr.update(z_lr)
        stats_hr.update(z_hr)
        host = []
        for t in (z_lr, z_hr, lr, hr):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            host.append(h)
        ev = torch.cuda.Event()
        ev.record()
        if pending is not None:
            write(*pending)
        pending = (list(batch['aoi']), host, ev)
    if pending is not None:
        write(*pending)
    return stats_lr, stats_hr


def write_pair_stats_json(stats_lr: RunningStats, stats_hr: RunningStats, path: str) -> None:
    """latent_stats.json schema of the reference (encode_latents.py:521-530)."""
    d = {'lr_latent': {k: v.tolist() for k, v in stats_lr.get_stats_dict().items()},
         'hr_latent': {k: v.tolist() for k, v in stats_hr.get_stats_dict().items()}}
    with open(path, 'w') as f:
        json.dump(d, f, indent=4)


def write_stats_json(stats: RunningStats, path: str) -> None:
    d = {k: v.tolist() for k, v in stats.get_stats_dict().items()}
    with open(path, 'w') as f:
        json.dump(d, f, indent=2)
