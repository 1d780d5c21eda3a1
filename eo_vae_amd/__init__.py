"""eo_vae_amd: MI355X-native (gfx950) implementation of the EO-VAE data-parallel hot path.

Hand-written HIP kernels behind a C-ABI (include/eovae_hip.h, csrc/), Python host code that
mirrors the reference's module tree and API (eo_vae.models.new_autoencoder.EOFluxVAE ...)."""

__version__ = '0.1.0'
