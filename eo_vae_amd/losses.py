"""Reconstruction / consistency loss on the HIP kernels
(reference: eo_vae/models/modules/consistency_loss.py:329-483).

On the hot path: pixel term (L1 or Charbonnier) and the scheduled MS-SSIM term, plus an
opt-in KL term that the reference's training path lacks (default weight 0
=> identical to the reference).  The scheduled spectral (SAM), spatial (gradient-difference)
and frequency (focal frequency, 1000-step warm-up) branches -- weight 0 in configs/eo-vae.yaml,
SURVEY.md 8f N4 -- run on their own HIP kernels too.  The DOFA feature branch needs a pretrained
third-party network and stays out of scope: a non-zero feature_weight raises."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class CharbonnierLoss(nn.Module):  # :12-21
    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps

    def forward(self, pred, target):
        return ops.charbonnier_loss(pred, target, self.eps)


class SSIMLoss(nn.Module):
    """1 - MS-SSIM(data_range=6, gaussian sigma 1.5 => 11 taps, 5 scales) (:24-37)."""

    def __init__(self, channels=12):
        super().__init__()
        self.data_range = 6.0
        self.betas = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)

    def forward(self, pred, target):
        return 1.0 - ops.msssim(pred, target, self.data_range, self.betas)


class SAMLoss(nn.Module):  # :186-209
    def __init__(self, eps=1e-8):
        super().__init__()
        self.eps = eps

    def forward(self, x_rec, x_true):
        return ops.sam_loss(x_rec, x_true, self.eps)


class GradientDifferenceLoss(nn.Module):  # :243-269
    def __init__(self, alpha=1.0):
        super().__init__()
        if alpha != 1.0:
            raise NotImplementedError('GradientDifferenceLoss: only alpha = 1 (the value EOConsistencyLoss uses)')
        self.alpha = alpha

    def forward(self, pred, target):
        return ops.gradient_difference_loss(pred, target)


class FocalFrequencyLoss(nn.Module):  # ffl.py:17-102
    def __init__(self, loss_weight=1.0, alpha=1.0, patch_factor=1, ave_spectrum=False, log_matrix=False,
                 batch_matrix=False):
        super().__init__()
        if ave_spectrum or not batch_matrix:
            raise NotImplementedError('FocalFrequencyLoss: only ave_spectrum=False, batch_matrix=True '
                                      '(the flags of consistency_loss.py:388-395) run on the HIP kernels')
        self.loss_weight, self.alpha, self.patch_factor = loss_weight, alpha, patch_factor
        self.log_matrix = log_matrix

    def forward(self, pred, target, matrix=None, **kwargs):
        if matrix is not None:
            raise NotImplementedError('FocalFrequencyLoss: a precomputed weight matrix is not supported')
        return ops.focal_frequency_loss(pred, target, self.alpha, self.patch_factor, self.log_matrix) * self.loss_weight


class EOConsistencyLoss(nn.Module):
    def __init__(self, pixel_weight: float = 1.0, rec_loss_type: str = 'l1',
                 spectral_weight: float = 0.0, spatial_weight: float = 0.0, freq_weight: float = 0.0,
                 feature_weight: float = 0.0, msssim_weight: float = 0.0,
                 spectral_start_step: int = 0, spatial_start_step: int = 0, freq_start_step: int = 0,
                 feature_start_step: int = 0, msssim_start_step: int = 0, patch_factor: int = 2,
                 ffl_alpha: float = 1.0, dofa_net: nn.Module = None, kl_weight: float = 0.0):
        super().__init__()
        if feature_weight > 0:
            raise NotImplementedError('feature_weight > 0: the DOFA semantic loss (consistency_loss.py:304-326) '
                                      'needs a pretrained DOFA network and is outside the MI355X hot path')
        self.rec_loss_type = rec_loss_type
        self.starts = {'spectral': spectral_start_step, 'spatial': spatial_start_step,
                       'freq': freq_start_step, 'feature': feature_start_step,
                       'msssim': msssim_start_step}
        self.weights = {'pixel': pixel_weight, 'spectral': spectral_weight,
                        'spatial': spatial_weight, 'freq': freq_weight, 'feature': feature_weight,
                        'msssim': msssim_weight, 'kl': kl_weight}
        self.char_loss = CharbonnierLoss()
        self.msssim_loss = SSIMLoss()
        self.sam_loss = SAMLoss()
        self.grad_loss = GradientDifferenceLoss()
        self.fft_loss = FocalFrequencyLoss(loss_weight=1.0, alpha=ffl_alpha, patch_factor=patch_factor,
                                           ave_spectrum=False, batch_matrix=True, log_matrix=True)  # :388-395

    def forward(self, inputs, wvs, reconstructions, global_step: int = 0, split: str = 'train',
                posterior=None, **kwargs):
        logs = {}
        total = torch.zeros((), dtype=torch.float32, device=inputs.device)
        if self.weights['pixel'] > 0:
            if self.rec_loss_type == 'l1':
                l_rec = ops.l1_loss(reconstructions, inputs)
            elif self.rec_loss_type == 'char':
                l_rec = self.char_loss(reconstructions, inputs)
            else:
                raise ValueError("rec_loss_type must be 'l1' or 'char'")
            total = total + self.weights['pixel'] * l_rec
            logs[f'{split}/loss_rec'] = l_rec.detach()
        if self.weights['spectral'] > 0 and global_step >= self.starts['spectral']:  # :423-428
            l_sam = self.sam_loss(reconstructions, inputs)
            total = total + self.weights['spectral'] * l_sam
            logs[f'{split}/loss_spectral'] = l_sam.detach()
        if self.weights['spatial'] > 0 and global_step >= self.starts['spatial']:  # :430-435
            l_spat = self.grad_loss(reconstructions, inputs)
            total = total + self.weights['spatial'] * l_spat
            logs[f'{split}/loss_spatial'] = l_spat.detach()
        if self.weights['freq'] > 0 and global_step >= self.starts['freq']:  # :437-464
            raw = self.fft_loss(reconstructions, inputs)
            warmup = min(1.0, max(0.0, (global_step - self.starts['freq']) / 1000))
            w_now = self.weights['freq'] * warmup
            total = total + raw * w_now
            logs[f'{split}/loss_freq_raw'] = raw.detach()
            logs[f'{split}/ffl_weight'] = torch.tensor(w_now)
        if self.weights['msssim'] > 0 and global_step >= self.starts['msssim']:
            l_ms = self.msssim_loss(reconstructions, inputs)
            total = total + self.weights['msssim'] * l_ms
            logs[f'{split}/loss_msssim'] = l_ms.detach()
        if self.weights['kl'] > 0 and posterior is not None:
            l_kl = posterior.kl().mean()
            total = total + self.weights['kl'] * l_kl
            logs[f'{split}/loss_kl'] = l_kl.detach()
        logs[f'{split}/loss_total'] = total.detach()
        return total, logs
