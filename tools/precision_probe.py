"""Whole-model error of each conv arithmetic mode vs the golden cfg1 fixture (reference fp32 CPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import eo_vae_cpu as O
from eo_vae_amd import ops
from tests.test_gpu_model import build_pair
g = dict(np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'full_cfg1.npz')))
_, prod = build_pair({}, 31)
prod.eval()
wvs = torch.tensor(O.WAVELENGTHS['S2RGB']).cuda()
x = O.synthetic_batch(2, 3, 256, seed=1234).cuda()
for prec in (sys.argv[1:] or ['f32', 'bf16x3', 'f16c8', 'bf16']):
    ops.CONV_PRECISION = prec
    rec = prod.reconstruct(x, wvs)[:, :, ::8, ::8].double().cpu()
    spat = prod.encode_spatial_normalized(x, wvs)[:, :, ::2, ::2].double().cpu()
    r0, s0 = torch.from_numpy(g['recon_sub']).double(), torch.from_numpy(g['spatial_sub']).double()
    print(f'{prec:7s} recon max-rel {((rec-r0).abs().max()/r0.abs().max()).item():.2e}  rms-rel {((rec-r0).norm()/r0.norm()).item():.2e} | '
          f'latent max-rel {((spat-s0).abs().max()/s0.abs().max()).item():.2e} rms-rel {((spat-s0).norm()/s0.norm()).item():.2e}')
