from eo_vae_amd.layers import AttnBlock, Downsample, ResnetBlock, Upsample  # noqa: F401
