"""Diagonal-Gaussian posterior (reference: eo_vae/models/modules/distributions.py:19-102)."""
from __future__ import annotations

import torch

from . import ops


class DiagonalGaussianDistribution:
    """moments [B, 2Z, H, W] = (mean | logvar); logvar clamped to [-30, 20].

    sample/kl run in fused HIP kernels; mean/logvar/std/var stay available as lazily
    computed tensors for API compatibility."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.deterministic = deterministic
        self.mean = torch.chunk(parameters, 2, dim=1)[0]
        self._lazy = {}

    def _get(self, name):
        if name not in self._lazy:
            logvar = torch.clamp(torch.chunk(self.parameters, 2, dim=1)[1], -30.0, 20.0)
            if self.deterministic:
                zero = torch.zeros_like(self.mean)
                self._lazy.update(logvar=logvar, std=zero, var=zero)
            else:
                self._lazy.update(logvar=logvar, std=torch.exp(0.5 * logvar), var=torch.exp(logvar))
        return self._lazy[name]

    logvar = property(lambda self: self._get('logvar'))
    std = property(lambda self: self._get('std'))
    var = property(lambda self: self._get('var'))

    def sample(self, noise: torch.Tensor = None) -> torch.Tensor:
        """mean + std * eps.  Like the reference (:44-46) the noise comes from the CPU default
        generator unless it is passed in explicitly (parity tests inject it)."""
        if self.deterministic:
            return self.mean
        if noise is None:
            noise = torch.randn(self.mean.shape)
        noise = noise.to(device=self.parameters.device, dtype=torch.float32, non_blocking=True)
        return ops.PosteriorSampleFn.apply(self.parameters, noise)

    def mode(self) -> torch.Tensor:
        return self.mean

    def kl(self, other=None) -> torch.Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        if other is None:
            return ops.PosteriorKLFn.apply(self.parameters)
        return 0.5 * torch.sum((self.mean - other.mean) ** 2 / other.var + self.var / other.var - 1.0
                               - self.logvar + other.logvar, dim=[1, 2, 3])

    def nll(self, sample, dims=(1, 2, 3)):
        if self.deterministic:
            return torch.Tensor([0.0])
        import math
        return 0.5 * torch.sum(math.log(2.0 * math.pi) + self.logvar
                               + (sample - self.mean) ** 2 / self.var, dim=list(dims))
