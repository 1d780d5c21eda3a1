from eo_vae_amd.losses import (CharbonnierLoss, EOConsistencyLoss, GradientDifferenceLoss, SAMLoss,  # noqa: F401
                               SSIMLoss)
