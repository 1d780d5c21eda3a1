"""`lightning` is not part of the MI355X image.  When it is importable EOFluxVAE subclasses the
real LightningModule (so `lightning.Trainer` from the reference's train.py drives it unchanged);
otherwise this module supplies the small slice of that API which EOFluxVAE.training_step uses
(manual optimisation) and eo_vae_amd.trainer.Trainer provides the fit loop."""
from __future__ import annotations

import torch

try:  # pragma: no cover - exercised only where lightning is installed
    from lightning import LightningModule as _RealLightningModule
    HAVE_LIGHTNING = True
except Exception:  # ImportError or a broken install
    _RealLightningModule = None
    HAVE_LIGHTNING = False


class _MiniLightningModule(torch.nn.Module):
    """Manual-optimisation subset: optimizers(), lr_schedulers(), manual_backward(), log_dict(),
    global_step, trainer attachment."""

    def __init__(self):
        super().__init__()
        self.automatic_optimization = True
        self._trainer = None
        self._optimizers, self._schedulers = [], []
        self._global_step = 0
        self.logged = {}

    # ---- wiring used by eo_vae_amd.trainer.Trainer ----
    def attach(self, trainer, optimizers, schedulers):
        self._trainer, self._optimizers, self._schedulers = trainer, list(optimizers), list(schedulers)

    @property
    def trainer(self):
        return self._trainer

    @property
    def global_step(self) -> int:
        return self._global_step

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device('cpu')

    def optimizers(self):
        return self._optimizers[0] if len(self._optimizers) == 1 else self._optimizers

    def lr_schedulers(self):
        if not self._schedulers:
            return None
        return self._schedulers[0] if len(self._schedulers) == 1 else self._schedulers

    def manual_backward(self, loss, *args, **kwargs):
        loss.backward(*args, **kwargs)
        reducer = getattr(self._trainer, 'reducer', None)
        if reducer is not None:  # wait for the bucketed RCCL all-reduces launched during backward
            reducer.finish()

    def log_dict(self, metrics, **kwargs):
        self.logged.update(metrics)
        if self._trainer is not None:
            self._trainer.log_metrics(metrics, **kwargs)

    def log(self, name, value, **kwargs):
        self.log_dict({name: value}, **kwargs)

    # hooks the fit loop may call
    def configure_optimizers(self):
        raise NotImplementedError

    def on_train_start(self):
        pass


LightningModule = _RealLightningModule if HAVE_LIGHTNING else _MiniLightningModule
