from eo_vae_amd.losses import FocalFrequencyLoss  # noqa: F401
