from eo_vae_amd.losses import CharbonnierLoss, EOConsistencyLoss, SSIMLoss  # noqa: F401
