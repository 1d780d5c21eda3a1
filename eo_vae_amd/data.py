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

    def __init__(self, modalities, batch_size, size, steps, seed, device=None, mode='random', prep=None, raw_size=None):
        self.modalities, self.batch_size, self.size = list(modalities), batch_size, size
        self.steps, self.seed, self.device, self.mode = steps, seed, device, mode
        self.prep, self.raw_size = prep, raw_size or size

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator('cpu').manual_seed(self.seed)
        rng = random.Random(self.seed)
        if self.prep is not None:
            # RAW-valued batches (sensor units: z-score-like fields pushed through the inverse of the legacy statistics), every
            # modality present as the TerraMesh loader delivers them; GpuBatchPrep picks one and prepares it on the GPU
            for _ in range(self.steps):
                raw = {}
                for mod in (self.modalities if self.mode == 'random' else [self.mode]):
                    st = NORM_STATS_LEGACY[mod]
                    z = synthetic_images(self.batch_size, len(st['mean']), self.raw_size, g)
                    raw[mod] = z * torch.tensor(st['std']).view(1, -1, 1, 1) + torch.tensor(st['mean']).view(1, -1, 1, 1)
                if len(self.prep.modalities) == 1:
                    raw = {'image': raw[self.prep.modalities[0]]}
                yield self.prep(raw)
            return
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
                 target_size=(256, 256), train_steps=2000, val_steps=4, seed=1234, normalize=True, norm_scheme='legacy',
                 gpu_batch_prep=None, raw_size=None, **ignored):
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
        # gpu_batch_prep: the loaders deliver RAW sensor values and the reference's collate (normalise, resize, D4, random
        # modality) runs as one HIP launch per batch (GpuBatchPrep) -- the datamodule arguments `normalize`, `norm_scheme`,
        # `target_size`, `train_collate_mode` mean what they mean in TerraMeshDataModule (terramesh_datamodule.py:600-680)
        self.gpu_batch_prep, self.normalize, self.norm_scheme = gpu_batch_prep, normalize, norm_scheme
        self.raw_size = raw_size

    def setup(self, stage=None, device=None, rank=0):
        self.device, self.rank = device, rank
        if self.gpu_batch_prep is None:      # default: on whenever the trainer hands over a GPU
            self.gpu_batch_prep = device is not None and torch.device(device).type == 'cuda'

    def _prep(self, mods, mode):
        if not self.gpu_batch_prep:
            return None
        return GpuBatchPrep(mods, self.normalize, self.norm_scheme, (self.size, self.size), mode, self.device,
                            random.Random(self.seed + 31 * self.rank + (0 if mode == 'train' else 7919)))

    def train_dataloader(self):
        mods = self.modalities if self.train_mode == 'random' else [self.train_mode]
        return SyntheticLoader(self.modalities, self.batch_size, self.size, self.train_steps,
                               self.seed + self.rank, self.device, self.train_mode, self._prep(mods, 'train'), self.raw_size)

    def val_dataloader(self):
        mode = self.val_mode if self.val_mode in WAVELENGTHS else 'random'
        mods = self.modalities if mode == 'random' else [mode]
        return SyntheticLoader(self.modalities, self.eval_batch_size, self.size, self.val_steps,
                               self.seed + 7919 + self.rank, self.device, mode, self._prep(mods, 'eval'), self.raw_size)


# ------------------------------------------------------------------------------------------------------------------
# N2: GPU batch preparation = the step in front of the hot path (terramesh_datamodule.py:130-339 normalisers, :347-369
# D4 augmentation, :476-479 resize, :418-503 random-modality collate), as ONE HIP launch per batch.
# The tables are the reference's statistics (data, not code): 'custom' scheme = clip to [0, 10000] + z-score
# (Sentinel2L2ANorm :141-183, Sentinel2L1CNorm :218-262), 'legacy' scheme = (x - mean) / (std + 1e-8) (:277-296, :54-118).
# ------------------------------------------------------------------------------------------------------------------
NORM_STATS_LEGACY = {
    'S2L2A': {'mean': [1375.648, 1489.600, 1709.087, 1831.752, 2186.075, 2794.358, 3008.528, 3096.780, 3155.180, 3169.651,
                       2415.761, 1838.622],
              'std': [2101.107, 2138.673, 2033.628, 2118.186, 2061.646, 1869.234, 1801.386, 1841.173, 1734.404, 1751.174,
                      1375.131, 1284.165]},
    'S1RTC': {'mean': [-10.793, -17.198], 'std': [4.278, 4.346]},
    'S2L1C': {'mean': [2475.625, 2260.839, 2143.561, 2230.225, 2445.427, 2992.950, 3257.843, 3171.695, 3440.958, 1567.433,
                       561.076, 2562.809, 1924.178],
              'std': [1761.905, 1804.267, 1661.263, 1932.020, 1918.007, 1812.421, 1795.179, 1734.280, 1780.039, 1082.531,
                      512.077, 1350.580, 1177.511]},
    'S2RGB': {'mean': [110.349, 99.507, 75.843], 'std': [69.905, 53.708, 53.378]},
    'DEM': {'mean': [651.663], 'std': [928.168]},
}
S2L2A_MEAN = [1718.9949, 1825.5669, 2043.5834, 2175.4543, 2522.9522, 3114.2216, 3323.3469, 3417.3660,
              3470.9655, 3489.4869, 2725.9735, 2152.0551]
S2L2A_STD = [2126.3409, 2140.1035, 2044.6618, 2125.3351, 2065.3251, 1874.4652, 1808.0426, 1839.0210,
             1737.9521, 1738.5136, 1456.5919, 1365.1743]
S2L1C_MEAN = [2424.2556, 2207.7019, 2098.2302, 2167.1584, 2382.3115, 2938.8499, 3204.8447, 3126.6599, 3389.0706,
              1580.1287, 572.5726, 2552.1208, 1917.9390]
S2L1C_STD = [1700.3824, 1731.5450, 1610.9904, 1833.5536, 1808.5067, 1694.4427, 1678.2327, 1625.7446, 1659.3112,
             1093.5255, 515.6395, 1300.8892, 1151.6169]
NORM_CUSTOM = {'S2L2A': (S2L2A_MEAN, S2L2A_STD), 'S2L1C': (S2L1C_MEAN, S2L1C_STD)}


def norm_table(modality: str, scheme: str = 'legacy'):
    """(mean, divisor, clamp) of the reference's NormalizerFactory.create(modality, scheme) (:308-327): the custom scheme
    exists for S2L2A / S2L1C and clips to [0, 10000]; everything else falls back to the legacy z-score, whose divisor is
    std + 1e-8 (added in fp32, as the reference does)."""
    if scheme == 'custom' and modality in NORM_CUSTOM:
        mean, std = NORM_CUSTOM[modality]
        return torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32), (0.0, 10000.0)
    if modality not in NORM_STATS_LEGACY:
        raise ValueError(f'Unknown modality {modality} for normalization.')
    st = NORM_STATS_LEGACY[modality]
    return torch.tensor(st['mean'], dtype=torch.float32), torch.tensor(st['std'], dtype=torch.float32) + 1e-8, None


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


class GpuBatchPrep:
    """The reference's `single_modality_collate_fn` (terramesh_datamodule.py:418-503) on the GPU: a RAW batch -- either
    {'image': [B, C, H, W]} for one modality or {modality: [B, C_m, H, W], ...} -- becomes the hot path's batch
    {'image', 'wvs', 'modality'} through ONE eovae_batch_prep launch (normalise, bilinear resize to target_size, D4
    augmentation in train mode).  The random draws are the reference's, in its order, from Python's `random` (or the
    `rng` given): the modality (`choice` over the modalities present, multi-modality batches only), then horizontal flip,
    vertical flip (`random() > 0.5` each) and the quarter turns (`randint(0, 3)`)."""

    def __init__(self, modalities, normalize=True, norm_scheme='legacy', target_size=(224, 224), mode='train',
                 device=None, rng=None):
        self.modalities = list(modalities)
        self.normalize, self.target_size, self.mode = normalize, target_size, mode
        self.rng = rng if rng is not None else random
        self.device = device
        self.tables = {m: norm_table(m, norm_scheme) for m in self.modalities} if normalize else {}
        self._dev_tables = {}
        self._dev_wvs = {}     # one wavelength tensor OBJECT per (modality, device): the generators memoise per tensor object

    def _table(self, modality, device):
        key = (modality, str(device))
        if key not in self._dev_tables:
            mean, div, clamp = self.tables[modality]
            self._dev_tables[key] = (mean.to(device), div.to(device), clamp)
        return self._dev_tables[key]

    def __call__(self, batch):
        if len(self.modalities) == 1:
            modality = self.modalities[0]
            if 'image' not in batch:
                raise ValueError(f"Expected 'image' key for single modality {modality}, but found: {batch.keys()}")
            images = batch['image']
        else:
            present = [m for m in self.modalities if m in batch]
            if not present:
                raise ValueError(f'None of the specified modalities {self.modalities} found in batch keys {batch.keys()}')
            modality = self.rng.choice(present)
            images = batch[modality]
        images = torch.as_tensor(images)
        device = self.device if self.device is not None else (images.device if images.is_cuda else torch.device('cuda'))
        images = images.to(device=device, dtype=torch.float32, non_blocking=True)
        mean = div = clamp = None
        if self.normalize:
            mean, div, clamp = self._table(modality, device)
        size = None
        if self.target_size is not None and tuple(images.shape[-2:]) != tuple(self.target_size):
            size = tuple(self.target_size)
        hf = vf = False
        k = 0
        if self.mode == 'train':       # apply_batch_augmentations (:347-369): three draws, in this order
            hf = self.rng.random() > 0.5
            vf = self.rng.random() > 0.5
            k = self.rng.randint(0, 3)
        out = prepare_batch(images, mean, div, clamp, size, hf, vf, k)
        # the same tensor object every batch: dynamic_conv.wavelength_embedding memoises on the object, and a
        # fresh tensor per batch never hit (11 launches per step stayed, 64 device tensors stayed pinned in the memo)
        wkey = (modality, str(device))
        wvs = self._dev_wvs.get(wkey)
        if wvs is None:
            wvs = self._dev_wvs[wkey] = torch.tensor(WAVELENGTHS[modality], dtype=torch.float32, device=device)
        return {'image': out, 'wvs': wvs, 'modality': modality}
