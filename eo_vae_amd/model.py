"""Encoder / Decoder of EOFluxVAE on the HIP kernels (reference: eo_vae/models/model.py).

Constructor signatures, attribute names and the parameter tree are the reference's
(model.py:68-78, 224-234): 363 state_dict keys, 95 514 145 parameters at configs/eo-vae.yaml."""
from __future__ import annotations

import torch
import torch.nn as nn

from .dynamic_conv import DynamicConv, DynamicConv_decoder
from .layers import AttnBlock, Conv2d, Downsample, GroupNorm, ResnetBlock, Upsample


def _split_dynamic_kwargs(kwargs):
    kw = dict(kwargs) if kwargs else {}
    if kw.pop('use_adain', False):
        raise NotImplementedError('use_adain=True (WavelengthConditioner/AdaIN) is outside the hot '
                                  'path: configs/eo-vae.yaml does not enable it')
    kw.pop('mode', None)
    return kw.pop('wv_planes', 128), kw.pop('inter_dim', 128), kw


class Encoder(nn.Module):
    def __init__(self, resolution: int, in_channels: int, ch: int, ch_mult: list, num_res_blocks: int,
                 z_channels: int, use_dynamic_ops: bool = False, dynamic_conv_kwargs: dict = None):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.z_channels = resolution, in_channels, z_channels
        self.use_dynamic_ops, self.use_adain, self.cond_dim = use_dynamic_ops, False, None
        if use_dynamic_ops:
            wv_planes, inter_dim, extra = _split_dynamic_kwargs(dynamic_conv_kwargs)
            self.conv_in = DynamicConv(wv_planes=wv_planes, inter_dim=inter_dim, kernel_size=3,
                                       stride=1, padding=1, embed_dim=ch, **extra)
        else:
            self.conv_in = Conv2d(in_channels, ch, 3)
        self.in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for level in range(self.num_resolutions):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * self.in_ch_mult[level], ch * ch_mult[level]
            for _ in range(num_res_blocks):
                stage.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if level != self.num_resolutions - 1:
                stage.downsample = Downsample(block_in)
            self.down.append(stage)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = GroupNorm(block_in)           # GN + swish fused (model.py:193-194)
        self.conv_out = Conv2d(block_in, 2 * z_channels, 3)
        self.quant_conv = Conv2d(2 * z_channels, 2 * z_channels, 1)

    def forward(self, x, wvs=None):
        if self.use_dynamic_ops:
            assert wvs is not None, 'wvs must be provided for Dynamic Encoder'
            h = self.conv_in(x, wvs)
        else:
            h = self.conv_in(x)
        for level, stage in enumerate(self.down):
            for block in stage.block:
                h = block(h)
            if level != self.num_resolutions - 1:
                h = stage.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.quant_conv(self.conv_out(self.norm_out(h, packed=True)))


class Decoder(nn.Module):
    def __init__(self, ch: int, out_ch: int, ch_mult: list, num_res_blocks: int, resolution: int,
                 z_channels: int, use_dynamic_ops: bool = False, dynamic_conv_kwargs: dict = None):
        super().__init__()
        self.post_quant_conv = Conv2d(z_channels, z_channels, 1)
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.z_channels, self.resolution = z_channels, resolution
        self.use_dynamic_ops, self.use_adain, self.cond_dim = use_dynamic_ops, False, None
        if use_dynamic_ops:
            wv_planes, inter_dim, extra = _split_dynamic_kwargs(dynamic_conv_kwargs)
        block_in = ch * ch_mult[-1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, 3)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for level in reversed(range(self.num_resolutions)):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[level]
            for _ in range(num_res_blocks + 1):
                stage.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if level != 0:
                stage.upsample = Upsample(block_in)
            self.up.insert(0, stage)
        self.norm_out = GroupNorm(block_in)
        if use_dynamic_ops:
            self.conv_out = DynamicConv_decoder(wv_planes=wv_planes, inter_dim=inter_dim,
                                                kernel_size=3, stride=1, padding=1,
                                                embed_dim=block_in, **extra)
        else:
            self.conv_out = Conv2d(block_in, out_ch, 3)

    def forward(self, z, wvs=None):
        if self.use_dynamic_ops and wvs is not None:
            self.conv_out.prefetch(wvs)   # output-kernel generator on a side stream, joined at conv_out
        h = self.conv_in(self.post_quant_conv(z))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for level in reversed(range(self.num_resolutions)):
            for block in self.up[level].block:
                h = block(h)
            if level != 0:
                h = self.up[level].upsample(h)
        h = self.norm_out(h, packed=True)     # consumed by conv_out only (plain or dynamic 3x3)
        if self.use_dynamic_ops:
            assert wvs is not None, 'wvs must be provided for Dynamic Decoder'
            return self.conv_out(h, wvs)
        return self.conv_out(h)
