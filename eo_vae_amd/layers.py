"""Building blocks of the Flux-style autoencoder on the HIP kernels.

Same module tree / parameter names as the reference (eo_vae/models/modules/layers.py) so that
state_dicts are interchangeable; every forward is a call into libeovae_hip.so (ops.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters/initialisation, HIP implicit-GEMM forward/backward.

    mode: 'normal' (stride 1, same padding), 'down' (zero pad right/bottom + stride 2,
    layers.py:33-37) or 'up' (nearest x2 folded into the tile load, layers.py:47-50)."""

    def __init__(self, cin, cout, kernel_size, mode='normal'):
        stride, padding = (2, 0) if mode == 'down' else (1, (kernel_size - 1) // 2)
        super().__init__(cin, cout, kernel_size, stride=stride, padding=padding)
        self.mode = mode

    def forward(self, x, residual=None):
        return ops.conv2d(x, self.weight, self.bias, residual, self.mode)


class GroupNorm(nn.GroupNorm):
    """GroupNorm(32, C, eps=1e-6, affine) optionally fused with the swish that follows it."""

    def __init__(self, channels, groups=32, silu=True):
        super().__init__(num_groups=groups, num_channels=channels, eps=1e-6, affine=True)
        self.silu = silu

    def forward(self, x, packed=False, cout=None):
        """packed=True: the caller feeds the result to a plain 3x3 Conv2d and nothing else; in f16c8 mode it is then
        produced as that kernel's operand records instead of fp32 (ops.mark_packed).  cout: that layer's output channels
        (MX operand records where the layer takes them, ops.records_format_for)."""
        return ops.group_norm_silu(x, self.weight, self.bias, self.num_groups, self.eps, self.silu, packed, cout)

    def forward_fork(self, x, packed=False, cout=None):
        """-> (normalised+activated x, x for the residual branch); see ops.GroupNormSiluForkFn."""
        return ops.group_norm_silu_fork(x, self.weight, self.bias, self.num_groups, self.eps, self.silu, packed, cout)


class Downsample(nn.Module):  # layers.py:25-37
    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = Conv2d(in_channels, in_channels, 3, mode='down')

    def forward(self, x):
        return self.conv(x)


class Upsample(nn.Module):  # layers.py:40-50
    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = Conv2d(in_channels, in_channels, 3, mode='up')

    def forward(self, x):
        return self.conv(x)


class ResnetBlock(nn.Module):
    """x + conv2(swish(GN(conv1(swish(GN(x)))))), 1x1 shortcut when Cin != Cout (layers.py:89-114).
    The residual addition rides in conv2's epilogue."""

    def __init__(self, in_channels: int, out_channels: int = None, cond_dim: int = None):
        super().__init__()
        if cond_dim is not None:
            raise NotImplementedError('AdaIN conditioning (use_adain) is outside the hot path')
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.cond_dim = in_channels, out_channels, None
        self.norm1 = GroupNorm(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, 3)
        self.norm2 = GroupNorm(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3)
        if in_channels != out_channels:
            self.nin_shortcut = Conv2d(in_channels, out_channels, 1)

    def forward(self, x, emb=None):
        # conv1 / conv2 are the only consumers of the two GroupNorm outputs: operand records
        a, xs = self.norm1.forward_fork(x, packed=True, cout=self.out_channels)
        h = self.conv1(a)
        skip = self.nin_shortcut(xs) if self.in_channels != self.out_channels else xs
        return self.conv2(self.norm2(h, packed=True, cout=self.out_channels), residual=skip)


class AttnBlock(nn.Module):
    """Single-head self-attention over H*W tokens, d = C (layers.py:117-142)."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.norm = GroupNorm(in_channels, silu=False)
        self.q = Conv2d(in_channels, in_channels, 1)
        self.k = Conv2d(in_channels, in_channels, 1)
        self.v = Conv2d(in_channels, in_channels, 1)
        self.proj_out = Conv2d(in_channels, in_channels, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        hn, x = self.norm.forward_fork(x)
        q = self.q(hn).view(b, c, h * w)
        k = self.k(hn).view(b, c, h * w)
        v = self.v(hn).view(b, c, h * w)
        o = ops.attention(q, k, v, float(c) ** -0.5).view(b, c, h, w)
        return self.proj_out(o, residual=x)
