"""`eo_vae.models.new_autoencoder` import path of the reference -> MI355X implementation."""
from eo_vae_amd.autoencoder import EOFluxVAE, get_cosine_schedule_with_warmup  # noqa: F401
from eo_vae_amd.distributions import DiagonalGaussianDistribution  # noqa: F401
from eo_vae_amd.model import Decoder, Encoder  # noqa: F401
