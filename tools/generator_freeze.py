"""TIMING EXPERIMENT, not product code (it used to sit behind an environment variable inside
eo_vae_amd/dynamic_conv.py): what the two weight generators cost a training step.

install() replaces the generators' forward by one that computes the generated kernels ONCE per (module, wavelength values)
and hands out the detached result: NO generator gradients, i.e. a step that trains nothing in the hyper-networks.  bench.py
calls it when EOVAE_TIMING_FREEZE_GENERATORS=1 and marks its JSON line `"timing_only"`; profiles/r04q_generator_cost.log is
the measurement it was written for."""
import torch


def install():
    from torch.nn.attention import SDPBackend, sdpa_kernel
    from eo_vae_amd import dynamic_conv as D
    frozen = {}

    def patch(cls):
        orig = cls._raw

        def _raw(self, wvs):
            key = (id(self), str(wvs.device), tuple(wvs.detach().reshape(-1).tolist()))
            hit = frozen.get(key)
            if hit is None:
                with torch.no_grad(), sdpa_kernel(SDPBackend.MATH):
                    hit = tuple(t.detach() for t in orig(self, wvs))
                frozen[key] = hit
            return hit
        cls._raw = _raw

    for name in dir(D):
        cls = getattr(D, name)
        if isinstance(cls, type) and '_raw' in cls.__dict__:
            patch(cls)
    print('[tools/generator_freeze] weight generators frozen: TIMING ONLY, no generator gradients', flush=True)
