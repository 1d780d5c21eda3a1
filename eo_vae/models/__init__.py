"""Drop-in import surface of the reference (eo_vae/models/__init__.py:1-12) backed by eo_vae_amd.

`eo_vae` stays a namespace package like the reference's (no top-level __init__.py, quirk q10).
The superseded wrappers AutoencoderKL / FluxAutoencoderKL are outside the hot path and are not
re-implemented; asking for them raises a descriptive ImportError."""
from eo_vae_amd.autoencoder import EOFluxVAE
from eo_vae_amd.model import Decoder, Encoder

__all__ = ('Encoder', 'Decoder', 'EOFluxVAE')


def __getattr__(name):
    if name in ('AutoencoderKL', 'FluxAutoencoderKL'):
        raise ImportError(f'eo_vae.models.{name} is superseded by EOFluxVAE (configs/eo-vae.yaml:12-13) '
                          'and is not part of the MI355X hot-path build')
    raise AttributeError(name)
