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


def full_traj_errors(prod, g=None):
    """THREE consecutive optimiser steps of the product at full size against tests/golden/full_cfg2_traj.npz, which
    gen_golden.py (full_traj) wrote by running the REFERENCE's step sequence (new_autoencoder.py:587-652: forward with the
    injected posterior noise, Charbonnier + 1e-6 KL, backward, clip_grad_norm_(1.0), Adam.step(); configure_optimizers
    :549-556) at a constant LR of 1e-5, B = 2, S2RGB 256x256, a new batch and draw per step.  `prod` = product EOFluxVAE on cuda,
    filled with det_fill(seed=31), built with EOConsistencyLoss(pixel_weight=1, rec_loss_type='char', kl_weight=1e-6),
    base_lr = the fixture's lr, no scheduler, clip_grad = 1.0.  The steps run through prod.training_step() itself (weight
    pre-pack, fused clip + Adam on the flat buffers).  Returns per step the relative errors of loss / Charbonnier / KL /
    pre-clip gradient norm, the third step's reconstruction error, and -- per probe tensor -- the error of the optimiser's
    UPDATE p3 - p0: of its norm, and of 256 sampled entries relative to the update's RMS entry."""
    if g is None:
        g = dict(np.load(os.path.join(GOLDEN, 'full_cfg2_traj.npz'), allow_pickle=False))
    prod.train()
    conf = prod.configure_optimizers()
    opts, scheds = conf if isinstance(conf, tuple) else (conf, [])
    assert not scheds and abs(opts[0].param_groups[0]['lr'] - float(g['lr'])) < 1e-12

    class Stub:
        reducer = None

        def log_metrics(self, *a, **k):
            pass
    prod.attach(Stub(), opts, [])
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB']).cuda()
    named = dict([('encoder.' + k, p) for k, p in prod.encoder.named_parameters()] +
                 [('decoder.' + k, p) for k, p in prod.decoder.named_parameters()])
    probes = [str(k) for k in g['probe_names']]
    p0 = {k: named[k].detach().clone() for k in probes}
    rel = lambda a, b: abs(float(torch.as_tensor(a).detach()) - float(b)) / abs(float(b))                    # noqa: E731
    out = {'loss_rel_err': [], 'charbonnier_rel_err': [], 'kl_rel_err': [], 'grad_norm_rel_err': []}
    rec_hook = {}
    fwd = prod.forward

    def capture(*a, **k):      # the third step's reconstruction, as training_step computes it
        r = fwd(*a, **k)
        rec_hook['rec'] = r[0].detach()
        return r
    prod.forward = capture
    try:
        for step in range(3):
            x = O.synthetic_batch(2, 3, 256, seed=1300 + step)
            assert abs(float(x.double().sum()) - float(g[f'x_checksum{step}'])) < 1e-6 * abs(float(g[f'x_checksum{step}'])) + 1e-6
            loss = prod.training_step({'image': x.cuda(), 'wvs': wvs}, step, noise=torch.from_numpy(g[f'eps{step}']).cuda())
            prod._global_step += 1
            torch.cuda.synchronize()
            out['loss_rel_err'].append(rel(loss, g[f'loss{step}']))
            out['charbonnier_rel_err'].append(rel(prod.logged['train/loss_rec'], g[f'charbonnier{step}']))
            out['kl_rel_err'].append(rel(prod.logged['train/loss_kl'], g[f'kl{step}']))
            out['grad_norm_rel_err'].append(rel(prod.logged['train/grad_norm'], g[f'gnorm{step}']))
    finally:
        prod.forward = fwd
    ref_r = torch.from_numpy(g['recon2_sub']).double()
    out['recon2_max_rel_err'] = float((rec_hook['rec'][:, :, ::8, ::8].double().cpu() - ref_r).abs().max() / ref_r.abs().max())
    k_s = g['delta_samples'].shape[1]
    worst_n = worst_s = 0.0
    worst_name = ''
    for name, n_ref, s_ref in zip(probes, g['delta_norms'], g['delta_samples']):
        d = (named[name].detach() - p0[name]).double()
        e_n = abs(float(d.norm()) - float(n_ref)) / float(n_ref)
        samp = d.reshape(-1)[torch.from_numpy(sample_index(d.numel(), k_s)).to(d.device)].cpu().numpy()
        rms = float(n_ref) / np.sqrt(d.numel())
        # sampled entries of the update: RMS error over the samples relative to the update's RMS entry (one entry of an
        # Adam update whose gradient sits at the rounding level can flip by 2 lr; the RMS over 256 samples cannot)
        e_s = float(np.sqrt(np.mean((samp - s_ref.astype(np.float64)) ** 2)) / rms)
        if max(e_n, e_s) > max(worst_n, worst_s):
            worst_name = name
        worst_n, worst_s = max(worst_n, e_n), max(worst_s, e_s)
    out.update(delta_worst_norm_rel_err=worst_n, delta_worst_sample_rms_rel_err=worst_s, delta_worst_tensor=worst_name,
               bn_mean_rel_err=float(np.abs(prod.bn.running_mean.cpu().numpy() - g['bn_running_mean']).max()
                                     / np.abs(g['bn_running_mean']).max()),
               bn_var_rel_err=float(np.abs(prod.bn.running_var.cpu().numpy() - g['bn_running_var']).max()
                                    / np.abs(g['bn_running_var']).max()))
    return out
