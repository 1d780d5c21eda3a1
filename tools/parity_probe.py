"""Full-size train-step parity (oracle/parity_checks.full_step_errors, the checker of tests/test_gpu_model.py and of bench.py's
parity leg) under combinations of the arithmetic options: which part of the scheme costs which error.
    python tools/parity_probe.py [case ...]        cases: full_cfg2_step (default), full_cfg3_step"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eo_vae_amd import lib as L, ops
from eo_vae_amd.autoencoder import EOFluxVAE
from eo_vae_amd.model import Decoder, Encoder
from oracle import eo_vae_cpu as O
from oracle import parity_checks as P

cases = sys.argv[1:] or ['full_cfg2_step']
lib = L.load()
cfg = O.model_cfg()
oracle = O.build_vae(cfg)
O.det_fill(oracle, seed=31)
prod = EOFluxVAE(Encoder(**cfg['encoder']), Decoder(**cfg['decoder']), torch.nn.Identity(), freeze_body=False)
state = {k: v.clone() for k, v in oracle.state_dict().items()}
prod.load_state_dict(state)
prod = prod.cuda()
del oracle
ops.CONV_PRECISION = 'f16c8'
for case in cases:
    # (mx, wgrad_corr, dgrad_mx, q16): r4 arithmetic ... the default (MX forward and data gradient, 16 x 16 kernels)
    for mx, corr, dmx, q16 in ((0, 1, 0, 0), (1, 1, 0, 0), (0, 0, 0, 0), (1, 0, 0, 0), (1, 0, 0, 1), (1, 0, 1, 1)):
        lib.eovae_set_option(b'c8_mx', mx)
        lib.eovae_set_option(b'c8_wgrad_corr', corr)
        lib.eovae_set_option(b'c8_dgrad_mx', dmx)
        lib.eovae_set_option(b'c8_q16', q16)
        prod.load_state_dict(state)      # a train-mode forward moves the latent BatchNorm's running statistics
        e = P.full_step_errors(prod, case=case)
        print(f"{case} mx={mx} wgrad_corr={corr} dgrad_mx={dmx} q16={q16}: loss {e['loss_rel_err']:.2e} recon {e['recon_max_rel_err']:.2e} "
              f"grad norm {e['grad_norm_rel_err']:.2e} worst tensor norm {e['grad_worst_norm_rel_err']:.2e} "
              f"worst sample {e['grad_worst_sample_rel_err']:.2e} ({e['grad_worst_tensor']})", flush=True)
