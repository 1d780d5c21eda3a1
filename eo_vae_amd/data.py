"""Batch contract of the hot path + a synthetic datamodule.

The reference's TerraMesh webdataset pipeline (eo_vae/datasets/terramesh*.py) is out of scope;
what the train step consumes is only the batch dict {'image': [B,C,H,W] fp32, 'wvs': [C] fp32,
'modality': str} (terramesh_datamodule.py:487) and the wavelength table (:18-50)."""
from __future__ import annotations

import random

import torch
import torch.nn.functional as F

WAVELENGTHS = {
    'S2RGB': [0.665, 0.56, 0.49],
    'S1RTC': [5.4, 5.6],
    'S2L2A': [0.443, 0.490, 0.560, 0.665, 0.705, 0.740, 0.783, 0.842, 0.865, 1.610, 2.190, 0.945],
    'S2L1C': [0.443, 0.490, 0.560, 0.665, 0.705, 0.740, 0.783, 0.842, 0.865, 0.945, 1.375, 1.610, 2.190],
}


def synthetic_images(batch: int, channels: int, size: int, generator: torch.Generator) -> torch.Tensor:
    """z-score-like, spatially correlated patches: low-res Gaussian field bilinearly
    upsampled x8 + 0.25 white noise, clamped to [-2.5, 8] (ranges of datasets/findings.md)."""
    low = torch.randn(batch, channels, max(size // 8, 1), max(size // 8, 1), generator=generator)
    x = F.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
    x = x + 0.25 * torch.randn(batch, channels, size, size, generator=generator)
    return x.clamp_(-2.5, 8.0)


class SyntheticLoader:
    """Endless-but-bounded iterator of synthetic batches; one modality per batch ('random' draws
    it per batch like train_collate_mode: random, terramesh_datamodule.py:418-503)."""

    def __init__(self, modalities, batch_size, size, steps, seed, device=None, mode='random'):
        self.modalities, self.batch_size, self.size = list(modalities), batch_size, size
        self.steps, self.seed, self.device, self.mode = steps, seed, device, mode

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator('cpu').manual_seed(self.seed)
        rng = random.Random(self.seed)
        for _ in range(self.steps):
            mod = rng.choice(self.modalities) if self.mode == 'random' else self.mode
            wvs = torch.tensor(WAVELENGTHS[mod], dtype=torch.float32)
            img = synthetic_images(self.batch_size, len(wvs), self.size, g)
            if self.device is not None:
                img = img.pin_memory().to(self.device, non_blocking=True) if img.device.type == 'cpu' and \
                    torch.cuda.is_available() else img.to(self.device)
                wvs = wvs.to(self.device)
            yield {'image': img, 'wvs': wvs, 'modality': mod}


class SyntheticDataModule:
    """Accepts the keyword arguments of the reference's TerraMeshDataModule config block and
    ignores the dataset-specific ones; unknown modalities (e.g. the stale 'S1GRD' of
    configs/eo-vae.yaml:63) are dropped with a note instead of raising."""

    def __init__(self, data_path=None, modalities=('S2L2A',), batch_size=16, eval_batch_size=None,
                 num_workers=0, train_collate_mode='random', val_collate_mode='S2L2A',
                 target_size=(256, 256), train_steps=2000, val_steps=4, seed=1234, **ignored):
        known = [m for m in modalities if m in WAVELENGTHS]
        dropped = [m for m in modalities if m not in WAVELENGTHS]
        if dropped:
            print(f'SyntheticDataModule: ignoring modalities without a wavelength table: {dropped}')
        self.modalities = known or ['S2L2A']
        self.batch_size, self.eval_batch_size = batch_size, eval_batch_size or batch_size
        self.size = int(target_size[0]) if isinstance(target_size, (list, tuple)) else int(target_size)
        self.train_mode, self.val_mode = train_collate_mode, val_collate_mode
        self.train_steps, self.val_steps, self.seed = train_steps, val_steps, seed
        self.device, self.rank = None, 0

    def setup(self, stage=None, device=None, rank=0):
        self.device, self.rank = device, rank

    def train_dataloader(self):
        return SyntheticLoader(self.modalities, self.batch_size, self.size, self.train_steps,
                               self.seed + self.rank, self.device, self.train_mode)

    def val_dataloader(self):
        mode = self.val_mode if self.val_mode in WAVELENGTHS else 'random'
        return SyntheticLoader(self.modalities, self.eval_batch_size, self.size, self.val_steps,
                               self.seed + 7919 + self.rank, self.device, mode)


# Sentinel-2 L2A statistics of the reference's custom normaliser (terramesh_datamodule.py:141-183)
S2L2A_MEAN = [1718.9949, 1825.5669, 2043.5834, 2175.4543, 2522.9522, 3114.2216, 3323.3469, 3417.3660,
              3470.9655, 3489.4869, 2725.9735, 2152.0551]
S2L2A_STD = [2126.3409, 2140.1035, 2044.6618, 2125.3351, 2065.3251, 1874.4652, 1808.0426, 1839.0210,
             1737.9521, 1738.5136, 1456.5919, 1365.1743]


def prepare_batch(raw: torch.Tensor, mean=None, std=None, clamp=None, target_size=None, hflip=False,
                  vflip=False, rot=0) -> torch.Tensor:
    """Fused GPU batch preparation (normalise -> bilinear resize -> D4 augmentation), one HIP
    kernel; the collate function of the reference does the same with separate host-side torch ops
    (terramesh_datamodule.py:463-482)."""
    from . import lib as L
    L.require_gpu(raw, 'prepare_batch')
    raw = raw.contiguous()
    n, c, hs, ws = raw.shape
    ht, wt = (hs, ws) if target_size is None else (int(target_size[0]), int(target_size[1]))
    hd, wd = (wt, ht) if rot % 2 else (ht, wt)
    out = torch.empty(n, c, hd, wd, dtype=torch.float32, device=raw.device)
    m = None if mean is None else torch.as_tensor(mean, dtype=torch.float32, device=raw.device).contiguous()
    s = None if std is None else torch.as_tensor(std, dtype=torch.float32, device=raw.device).contiguous()
    lo, hi = (1.0, 0.0) if clamp is None else (float(clamp[0]), float(clamp[1]))
    L.check(L.load().eovae_batch_prep(L.ptr(raw), L.ptr(m), L.ptr(s), L.ptr(out), n, c, hs, ws, ht, wt, lo, hi,
                                      int(hflip), int(vflip), int(rot) % 4, L.stream()), 'eovae_batch_prep')
    return out
