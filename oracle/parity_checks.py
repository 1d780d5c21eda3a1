"""TEST INFRASTRUCTURE (checker side), like everything under oracle/: comparisons of the HIP product path with
the committed reference fixtures that more than one checker runs -- tests/ and bench.py's `parity` leg (after the
timed region).  Never imported by the product (eo_vae_amd/)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import eo_vae_cpu as O

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def sample_index(numel, k=16):
    """tests/golden/gen_golden.py:sample_index -- the k flat indices stored per gradient tensor."""
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


FULL_STEP_CASES = {   # fixture -> (modality, seed of the synthetic batch); tests/golden/gen_golden.py full_step / full_step12
    'full_cfg2_step': ('S2RGB', 1234),
    'full_cfg3_step': ('S2L2A', 1236),
}


def full_step_errors(prod, g=None, case='full_cfg2_step'):
    """One full-size train-mode forward + backward of the product model `prod` (on cuda, filled with
    det_fill(seed=31)) against tests/golden/<case>.npz, which gen_golden.py wrote by running the
    REFERENCE (new_autoencoder.py:587-690: forward with the injected posterior noise, Charbonnier + 1e-6 KL,
    backward; B=2, 256x256; `full_cfg2_step` = S2RGB, `full_cfg3_step` = S2L2A with its 12-channel dynamic
    convolutions).  Returns relative errors of the loss scalars, the reconstruction, the global
    gradient norm and -- per parameter tensor -- the gradient's L2 norm and 16 sampled entries."""
    from eo_vae_amd import ops
    if g is None:
        g = dict(np.load(os.path.join(GOLDEN, case + '.npz'), allow_pickle=False))
    modality, seed = FULL_STEP_CASES[case]
    prod.train()
    wvs = torch.tensor(O.WAVELENGTHS[modality]).cuda()
    x = O.synthetic_batch(2, len(O.WAVELENGTHS[modality]), 256, seed=seed)
    assert abs(float(x.double().sum()) - float(g['x_checksum'])) < 1e-6 * abs(float(g['x_checksum'])) + 1e-6
    x = x.cuda()
    for p in prod.parameters():
        p.grad = None
    rec, post = prod(x, wvs, noise=torch.from_numpy(g['eps']).cuda())
    char = ops.charbonnier_loss(rec, x)
    l1 = ops.l1_loss(rec.detach(), x)
    kl = post.kl()
    loss = char + 1e-6 * kl.mean()
    loss.backward()
    torch.cuda.synchronize()
    rel = lambda a, b: abs(float(torch.as_tensor(a).detach()) - float(b)) / abs(float(b))                    # noqa: E731
    ref_r = torch.from_numpy(g['recon_sub']).double()
    out = {
        'loss_rel_err': rel(loss, g['loss']), 'charbonnier_rel_err': rel(char, g['charbonnier']),
        'l1_rel_err': rel(l1, g['l1']),
        'kl_max_rel_err': float((kl.detach().double().cpu() - torch.from_numpy(g['kl']).double()).abs().max()
                                / torch.from_numpy(g['kl']).double().abs().max()),
        'recon_max_rel_err': float((rec.detach()[:, :, ::8, ::8].double().cpu() - ref_r).abs().max() / ref_r.abs().max()),
    }
    named = dict([('encoder.' + k, p) for k, p in prod.encoder.named_parameters()] +
                 [('decoder.' + k, p) for k, p in prod.decoder.named_parameters()])
    tot = float(g['grad_norm'])
    sq = 0.0
    worst_norm = worst_samp = 0.0
    worst_name = ''
    for name, n_ref, s_ref in zip(g['grad_names'], g['grad_norms'], g['grad_samples']):
        gr = named[str(name)].grad.double()
        n = float(gr.norm())
        sq += n * n
        floor = max(float(n_ref), 1e-3 * tot)          # tensors whose true gradient is ~0 are pure round-off
        e_n = abs(n - float(n_ref)) / floor
        samp = gr.reshape(-1)[torch.from_numpy(sample_index(gr.numel())).to(gr.device)].cpu().numpy()
        # sampled entries against the tensor's RMS entry (an entry near zero has no relative error of its own)
        rms = floor / np.sqrt(gr.numel())
        e_s = float(np.abs(samp - s_ref.astype(np.float64)).max() / max(rms, np.abs(s_ref).max()))
        if max(e_n, e_s) > max(worst_norm, worst_samp):
            worst_name = str(name)
        worst_norm, worst_samp = max(worst_norm, e_n), max(worst_samp, e_s)
    out.update(grad_norm_rel_err=abs(np.sqrt(sq) - tot) / tot, grad_worst_norm_rel_err=worst_norm,
               grad_worst_sample_rel_err=worst_samp, grad_worst_tensor=worst_name,
               bn_mean_rel_err=float(np.abs(prod.bn.running_mean.cpu().numpy() - g['bn_running_mean']).max()
                                     / np.abs(g['bn_running_mean']).max()),
               bn_var_rel_err=float(np.abs(prod.bn.running_var.cpu().numpy() - g['bn_running_var']).max()
                                    / np.abs(g['bn_running_var']).max()))
    for p in prod.parameters():
        p.grad = None
    return out


def full_encode_errors(prod, g=None):
    """cfg5's path at full size against tests/golden/full_cfg5_encode.npz (gen_golden.py full_encode: the REFERENCE's
    encode_spatial_normalized, new_autoencoder.py:480-502, on two S2L2A batches of 2 x 12 x 256 x 256, then
    RunningStatsButFast, encode_latents.py:65-94).  `prod` = product model on cuda filled with det_fill(seed=31)."""
    from eo_vae_amd.encode_latents import RunningStats
    if g is None:
        g = dict(np.load(os.path.join(GOLDEN, 'full_cfg5_encode.npz'), allow_pickle=False))
    prod.eval()
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A']).cuda()
    stats = RunningStats(32, device='cuda')
    out = {}
    for i in range(2):
        x = O.synthetic_batch(2, 12, 256, seed=1240 + i)
        assert abs(float(x.double().sum()) - float(g[f'x_checksum{i}'])) < 1e-6 * abs(float(g[f'x_checksum{i}'])) + 1e-6
        with torch.no_grad():
            z = prod.encode_spatial_normalized(x.cuda(), wvs)
        stats.update(z)
        ref = torch.from_numpy(g[f'z_sub{i}']).double()
        out[f'latent{i}_max_rel_err'] = float((z[:, :, ::2, ::2].double().cpu() - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    sd = stats.get_stats_dict()
    for k in ['mean', 'var', 'std', 'min', 'max', 'count']:
        mine = torch.as_tensor(sd[k]).detach().double().cpu().reshape(-1).numpy()
        ref = g['stats_' + k].astype(np.float64).reshape(-1)
        out[f'stats_{k}_rel_err'] = float(np.abs(mine - ref).max() / max(np.abs(ref).max(), 1e-30))
    return out
