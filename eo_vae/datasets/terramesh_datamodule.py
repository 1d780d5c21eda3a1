"""Batch-contract stand-in for the reference's TerraMeshDataModule `_target_` (configs/eo-vae.yaml:61):
the webdataset pipeline is out of scope, the hot path is driven by synthetic batches."""
from eo_vae_amd.data import WAVELENGTHS, SyntheticDataModule as TerraMeshDataModule  # noqa: F401
