"""`eo_vae.models.modules` of the reference: only the hot-path members exist here (the GAN / LPIPS
losses the reference's __init__ eagerly imports need uninstalled third-party packages)."""
from eo_vae_amd.distributions import DiagonalGaussianDistribution
from eo_vae_amd.losses import EOConsistencyLoss

__all__ = ('DiagonalGaussianDistribution', 'EOConsistencyLoss')
