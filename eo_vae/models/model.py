"""`eo_vae.models.model` import path of the reference -> MI355X implementation."""
from eo_vae_amd.model import Decoder, Encoder  # noqa: F401
