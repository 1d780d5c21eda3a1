from eo_vae_amd.distributions import DiagonalGaussianDistribution  # noqa: F401
