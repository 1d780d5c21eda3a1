#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What it does
  * imports the reference's model code through synthetic parent packages (the
    reference's own ``__init__`` files eagerly import uninstalled dependencies;
    ``lightning`` / ``omegaconf`` get two in-process stand-in modules that the
    arithmetic never touches) -- SURVEY.md section 8(c);
  * fills reference modules with ``oracle.eo_vae_cpu.det_fill`` (closed form,
    no weight files), runs them on seeded inputs and stores inputs + outputs
    as small ``.npz`` fixtures (data only -- no reference source is stored);
  * as a by-product asserts that the oracle restatement agrees with the
    reference on every fixture (so a broken oracle can never be committed).

MS-SSIM has no reference-side fixture: ``torchmetrics`` is not installed
("parity unpinned").
"""

from __future__ import annotations

import ast
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import eo_vae_cpu as O  # noqa: E402


def import_reference():
    for name, rel in [('eo_vae', 'eo_vae'), ('eo_vae.models', 'eo_vae/models'),
                      ('eo_vae.models.modules', 'eo_vae/models/modules')]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = m
    lt = types.ModuleType('lightning')
    lt.LightningModule = torch.nn.Module
    sys.modules['lightning'] = lt
    oc = types.ModuleType('omegaconf')
    oc.OmegaConf = object
    sys.modules['omegaconf'] = oc
    R = types.SimpleNamespace()
    R.model = importlib.import_module('eo_vae.models.model')
    R.layers = importlib.import_module('eo_vae.models.modules.layers')
    R.dyn = importlib.import_module('eo_vae.models.modules.dynamic_conv')
    R.dist = importlib.import_module('eo_vae.models.modules.distributions')
    R.vae = importlib.import_module('eo_vae.models.new_autoencoder')
    return R


def ref_class_from_source(path, cls_name, ns):
    """Execute ONE class definition of a reference script that cannot be imported
    as a whole (top-level imports of uninstalled packages)."""
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    mod = ast.Module(body=[node], type_ignores=[])
    exec(compile(mod, path, 'exec'), ns)
    return ns[cls_name]


def ref_defs_from_source(path, names, ns):
    """Execute selected top-level class / function definitions of a reference script that cannot be
    imported as a whole (same reason as ref_class_from_source)."""
    tree = ast.parse(open(path).read())

    def name_of(n):
        if isinstance(n, (ast.ClassDef, ast.FunctionDef)):
            return n.name
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):
            return n.targets[0].id      # module-level tables (WAVELENGTHS, NORM_STATS_LEGACY)
        return None
    nodes = [n for n in tree.body if name_of(n) in names]
    assert sorted(name_of(n) for n in nodes) == sorted(names), [name_of(n) for n in nodes]
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, 'exec'), ns)
    return [ns[n] for n in names]


def npy(t):
    return t.detach().cpu().numpy().copy()   # a copy: running statistics are updated in place later


def close(a, b, tol, what):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-30
    assert err <= tol * max(ref, 1.0), f'oracle != reference for {what}: {err} (ref max {ref})'
    return err


def hashed(shape, salt, scale=1.0):
    n = int(np.prod(shape))
    return (O._hash_uniform(n, salt).reshape(shape) * (scale * 12 ** 0.5)).float()


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f'  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)')


def grads_of(module, out, gy, x=None):
    module.zero_grad()
    if x is not None and x.grad is not None:
        x.grad = None
    out.backward(gy)
    g = {k: npy(p.grad) for k, p in module.named_parameters()}
    return g


def per_op(R):
    print('per-op fixtures')
    # ---- ResnetBlock (with and without nin_shortcut), layers.py:53-114
    for tag, cin, cout, hw in [('res_same', 32, 32, 16), ('res_proj', 32, 64, 16)]:
        ref = R.layers.ResnetBlock(cin, cout)
        mine = O.ResnetBlock(cin, cout)
        O.det_fill(ref, seed=11)
        mine.load_state_dict(ref.state_dict())
        x = hashed((2, cin, hw, hw), 101, 1.0).requires_grad_(True)
        gy = hashed((2, cout, hw, hw), 102, 1.0)
        y = ref(x)
        gp = grads_of(ref, y, gy, x)
        gx = npy(x.grad)
        x2 = x.detach().clone().requires_grad_(True)
        y2 = mine(x2)
        gp2 = grads_of(mine, y2, gy)
        close(y2, y, 1e-5, tag + '.y')
        close(x2.grad, gx, 1e-4, tag + '.gx')
        for k in gp:
            close(gp2[k], gp[k], 1e-3, f'{tag}.g[{k}]')  # k.bias grad is analytically 0: pure round-off
        save('op_' + tag, x=npy(x), gy=npy(gy), y=npy(y), gx=gx,
             **{'g.' + k: v for k, v in gp.items()})
    # ---- AttnBlock, layers.py:117-142
    for tag, c, hw in [('attn_8', 64, 8), ('attn_16', 64, 16)]:
        ref, mine = R.layers.AttnBlock(c), O.AttnBlock(c)
        O.det_fill(ref, seed=12, gain=2.0)
        mine.load_state_dict(ref.state_dict())
        x = hashed((2, c, hw, hw), 111, 1.0).requires_grad_(True)
        gy = hashed((2, c, hw, hw), 112, 1.0)
        y = ref(x)
        gp = grads_of(ref, y, gy, x)
        gx = npy(x.grad)
        x2 = x.detach().clone().requires_grad_(True)
        y2 = mine(x2)
        gp2 = grads_of(mine, y2, gy)
        close(y2, y, 1e-5, tag + '.y')
        close(x2.grad, gx, 1e-4, tag + '.gx')
        for k in gp:
            close(gp2[k], gp[k], 1e-3, f'{tag}.g[{k}]')  # k.bias grad is analytically 0: pure round-off
        save('op_' + tag, x=npy(x), gy=npy(gy), y=npy(y), gx=gx,
             **{'g.' + k: v for k, v in gp.items()})
    # ---- Downsample / Upsample, layers.py:25-50
    for tag, cls_r, cls_o, hw in [('down', R.layers.Downsample, O.Downsample, 16),
                                  ('up', R.layers.Upsample, O.Upsample, 8)]:
        ref, mine = cls_r(32), cls_o(32)
        O.det_fill(ref, seed=13)
        mine.load_state_dict(ref.state_dict())
        x = hashed((2, 32, hw, hw), 121, 1.0).requires_grad_(True)
        y = ref(x)
        gy = hashed(tuple(y.shape), 122, 1.0)
        gp = grads_of(ref, y, gy, x)
        gx = npy(x.grad)
        x2 = x.detach().clone().requires_grad_(True)
        y2 = mine(x2)
        gp2 = grads_of(mine, y2, gy)
        close(y2, y, 1e-5, tag + '.y')
        close(x2.grad, gx, 1e-4, tag + '.gx')
        for k in gp:
            close(gp2[k], gp[k], 1e-3, f'{tag}.g[{k}]')  # k.bias grad is analytically 0: pure round-off
        save('op_' + tag, x=npy(x), gy=npy(gy), y=npy(y), gx=gx,
             **{'g.' + k: v for k, v in gp.items()})
    # ---- dynamic convs at FULL hyper-network size (wv_planes 256, 4 layers), dynamic_conv.py
    ref_in = R.dyn.DynamicConv(wv_planes=256, inter_dim=128, kernel_size=3, stride=1, padding=1,
                               embed_dim=128, num_layers=4)
    ref_out = R.dyn.DynamicConv_decoder(wv_planes=256, inter_dim=128, kernel_size=3, stride=1,
                                        padding=1, embed_dim=128, num_layers=4)
    my_in = O.DynamicConv(wv_planes=256, embed_dim=128, num_layers=4)
    my_out = O.DynamicConv_decoder(wv_planes=256, embed_dim=128, num_layers=4)
    O.det_fill(ref_in, seed=14)
    O.det_fill(ref_out, seed=15)
    my_in.load_state_dict(ref_in.state_dict())
    my_out.load_state_dict(ref_out.state_dict())
    ref_in.eval(), ref_out.eval(), my_in.eval(), my_out.eval()
    arrays = {}
    for mod, wl in O.WAVELENGTHS.items():
        wvs = torch.tensor(wl)
        c = len(wl)
        x = hashed((1, c, 12, 12), 131 + c, 1.0)
        h = hashed((1, 128, 12, 12), 141 + c, 1.0)
        with torch.no_grad():
            y_in = ref_in(x, wvs)
            y_out = ref_out(h, wvs)
            w_in, b_in = ref_in.get_distillation_weight(wvs)  # same tensors forward() convolves with
            w_out = ref_out.weight
            m_w_in, m_b_in = my_in.generate(wvs)
            m_w_out, m_b_out = my_out.generate(wvs)
            close(m_w_in, w_in, 1e-5, f'dyn_in.w[{mod}]')
            close(m_b_in, b_in, 1e-5, f'dyn_in.b[{mod}]')
            close(m_w_out, w_out, 1e-5, f'dyn_out.w[{mod}]')
            close(my_in(x, wvs), y_in, 1e-5, f'dyn_in.y[{mod}]')
            close(my_out(h, wvs), y_out, 1e-5, f'dyn_out.y[{mod}]')
        # decoder bias (scaled twice): recover from a zero input
        with torch.no_grad():
            b_out = ref_out(torch.zeros(1, 128, 4, 4), wvs)[0, :, 0, 0]
            close(m_b_out, b_out, 1e-5, f'dyn_out.b[{mod}]')
        arrays.update({f'{mod}.w_in': npy(w_in), f'{mod}.b_in': npy(b_in),
                       f'{mod}.w_out': npy(w_out), f'{mod}.b_out': npy(b_out),
                       f'{mod}.x': npy(x), f'{mod}.h': npy(h),
                       f'{mod}.y_in': npy(y_in), f'{mod}.y_out': npy(y_out)})
    save('op_dynconv', **arrays)
    # ---- posterior, distributions.py:19-102
    moments = hashed((2, 64, 8, 8), 151, 2.0)
    moments[0, 32:, 0, 0] = 40.0   # exercise the clamp(-30, 20)
    moments[1, 32:, 1, 1] = -50.0
    eps = hashed((2, 32, 8, 8), 152, 1.0)
    ref_p = R.dist.DiagonalGaussianDistribution(moments)
    my_p = O.DiagonalGaussianDistribution(moments)
    sample = ref_p.mean + ref_p.std * eps  # distributions.py:44-46 with the noise injected
    close(my_p.sample(eps), sample, 1e-6, 'posterior.sample')
    close(my_p.kl(), ref_p.kl(), 1e-6, 'posterior.kl')
    save('op_posterior', moments=npy(moments), eps=npy(eps), sample=npy(sample),
         kl=npy(ref_p.kl()), mode=npy(ref_p.mode()))


def whole_models(R):
    print('tiny whole-model fixtures')
    cfg = O.model_cfg(**O.TINY)
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    mine = O.build_vae(cfg)
    O.det_fill(ref, seed=21)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    mine.load_state_dict(ref.state_dict())
    out = {}
    for mod in ['S2RGB', 'S1RTC', 'S2L2A']:
        wvs = torch.tensor(O.WAVELENGTHS[mod])
        x = O.synthetic_batch(2, len(wvs), 64, seed=1234 + len(wvs))
        ref.eval(), mine.eval()
        with torch.no_grad():
            rec = ref.reconstruct(x, wvs)
            lat = ref.encode_to_latent(x, wvs)
            spat = ref.encode_spatial_normalized(x, wvs)
            mom = ref.encode(x, wvs).parameters
            dec = ref.decode_spatial_normalized(spat, wvs)
            close(mine.reconstruct(x, wvs), rec, 1e-4, f'tiny.recon[{mod}]')
            close(mine.encode_to_latent(x, wvs), lat, 1e-4, f'tiny.latent[{mod}]')
            close(mine.encode_spatial_normalized(x, wvs), spat, 1e-4, f'tiny.spatial[{mod}]')
            close(mine.decode_spatial_normalized(spat, wvs), dec, 1e-4, f'tiny.dec[{mod}]')
        out.update({f'{mod}.x': npy(x), f'{mod}.recon': npy(rec), f'{mod}.latent': npy(lat),
                    f'{mod}.spatial': npy(spat), f'{mod}.moments': npy(mom), f'{mod}.dec': npy(dec)})
    save('tiny_eval', **out)

    # ---- train-mode forward/backward + 3 optimiser steps (L1, clip 1.0, warmup schedule)
    print('tiny train trajectory')
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
    ref.train(), mine.train()
    base_lr, final_lr = 1e-3, 2e-4
    params_r = [p for p in ref.encoder.parameters()] + [p for p in ref.decoder.parameters()]
    opt_r = torch.optim.Adam(params_r, lr=base_lr)
    # new_autoencoder.py:568-581 with steps_per_epoch shrunk so 3 steps cross the warm-up
    sch_r = R.vae.get_cosine_schedule_with_warmup(opt_r, num_warmup_steps=2, num_training_steps=6,
                                                  base_lr=base_lr, final_lr=final_lr)
    st = O.TrainState(mine, base_lr=base_lr, final_lr=final_lr, warmup_epochs=1, decay_end_epoch=3,
                      clip_grad=1.0, steps_per_epoch=2)
    loss_o = O.EOConsistencyLoss(pixel_weight=1.0, rec_loss_type='l1')
    traj = {}
    for step in range(3):
        x = O.synthetic_batch(2, len(wvs), 64, seed=500 + step)
        eps = hashed((2, 32, 8, 8), 600 + step, 1.0)
        # reference forward with the posterior noise injected (distributions.py:44-46)
        torch.manual_seed(0)
        post = ref.encode(x, wvs)
        z = post.mean + post.std * eps
        from einops import rearrange
        zs = rearrange(z, '... c (i pi) (j pj) -> ... (c pi pj) i j', pi=2, pj=2)
        recon = ref.decode(ref._normalize_latent(zs), wvs)
        opt_r.zero_grad()
        loss = torch.nn.functional.l1_loss(recon, x)   # consistency_loss.py:418
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(params_r, 1.0)
        lr = opt_r.param_groups[0]['lr']
        if step == 0:
            names = [k for k, _ in list(ref.encoder.named_parameters())]
            gnorms = {('encoder.' + k): p.grad.norm().item() for k, p in ref.encoder.named_parameters()}
            gnorms.update({('decoder.' + k): p.grad.norm().item() for k, p in ref.decoder.named_parameters()})
            traj['grad_names'] = np.array(list(gnorms.keys()))
            traj['grad_norms'] = np.array(list(gnorms.values()), dtype=np.float64)
            traj['recon0'] = npy(recon)
            traj['kl0'] = npy(post.kl())
        opt_r.step()
        sch_r.step()
        l2, logs, gn2 = st.step(loss_o, x, wvs, noise=eps)
        close(l2, loss, 2e-5, f'traj.loss[{step}]')
        close(gn2, gn, 1e-3, f'traj.gnorm[{step}]')
        assert abs(logs['train/lr'] - lr) < 1e-12, (logs['train/lr'], lr)
        traj[f'x{step}'] = npy(x)
        traj[f'eps{step}'] = npy(eps)
        traj[f'loss{step}'] = np.float64(loss.item())
        traj[f'gnorm{step}'] = np.float64(gn.item())
        traj[f'lr{step}'] = np.float64(lr)
    sd_r, sd_m = ref.state_dict(), mine.state_dict()
    for k in sd_r:
        # Adam turns round-off-only gradients (e.g. a conv bias feeding a GroupNorm, whose true
        # gradient is 0) into +-lr steps, so parameters agree to ~sum(lr), not to 1e-4
        close(sd_m[k].float(), sd_r[k].float(), 2e-3, f'traj.param[{k}]')
    traj['bn_running_mean'] = npy(sd_r['bn.running_mean'])
    traj['bn_running_var'] = npy(sd_r['bn.running_var'])
    traj['bn_batches'] = np.int64(sd_r['bn.num_batches_tracked'].item())
    probe = ['encoder.down.0.block.0.conv1.weight', 'decoder.up.1.block.0.nin_shortcut.weight',
             'decoder.conv_out.weight_generator.fc_weight.weight', 'encoder.mid.attn_1.q.weight']
    for k in probe:
        traj['final.' + k] = npy(sd_r[k]).reshape(-1)[:64]
    save('tiny_train', **traj)


def full_size(R):
    print('full-size cfg1 fixture (B=2, S2RGB, 256x256) -- takes ~20 s')
    cfg = O.model_cfg()
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    O.det_fill(ref, seed=31)
    ref.eval()
    n_params = sum(p.numel() for p in ref.encoder.parameters()) + sum(p.numel() for p in ref.decoder.parameters())
    assert n_params == 95514145, n_params
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    x = O.synthetic_batch(2, 3, 256, seed=1234)
    with torch.no_grad():
        rec = ref.reconstruct(x, wvs)
        spat = ref.encode_spatial_normalized(x, wvs)
    mine = O.build_vae(cfg)
    mine.load_state_dict(ref.state_dict())
    mine.eval()
    with torch.no_grad():
        close(mine.reconstruct(x, wvs), rec, 1e-4, 'full.recon')
        close(mine.encode_spatial_normalized(x, wvs), spat, 1e-4, 'full.spatial')
    save('full_cfg1', x_checksum=np.float64(x.double().sum().item()),
         recon_sub=npy(rec[:, :, ::8, ::8]), spatial_sub=npy(spat[:, :, ::2, ::2]),
         recon_stats=np.array([[rec[:, c].mean().item(), rec[:, c].std().item(),
                                rec[:, c].min().item(), rec[:, c].max().item()] for c in range(3)]),
         n_params=np.int64(n_params), n_keys=np.int64(len(ref.state_dict())),
         keys=np.array(list(ref.state_dict().keys())))


def sample_index(numel, k=16):
    """k deterministic flat indices of a tensor of `numel` elements (both sides regenerate them)."""
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


def full_step(R):
    """The path bench.py times, at FULL size, on the reference itself: new_autoencoder.py:587-690 (train-mode
    forward with the posterior noise injected, loss, backward) on the configs/eo-vae.yaml model, B=2, S2RGB
    256x256, Charbonnier reconstruction loss (the reference config's rec_loss_type; the L1 value is stored
    beside it).  Stored: loss scalars, a reconstruction subsample, the global gradient norm, and for EVERY
    parameter its gradient's L2 norm + 16 sampled entries (SURVEY 8(c)(2) style)."""
    print('full-size cfg2 train-step fixture (B=2, S2RGB, 256x256, forward + backward) -- takes ~2 min')
    from einops import rearrange
    F = torch.nn.functional
    cfg = O.model_cfg()
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    O.det_fill(ref, seed=31)
    ref.train()
    ns = {'torch': torch, 'nn': torch.nn, 'F': F}
    Charb = ref_class_from_source(os.path.join(REF, 'eo_vae/models/modules/consistency_loss.py'),
                                  'CharbonnierLoss', ns)
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    x = O.synthetic_batch(2, 3, 256, seed=1234)
    eps = hashed((2, 32, 32, 32), 1600, 1.0)
    post = ref.encode(x, wvs)
    z = post.mean + post.std * eps                       # distributions.py:44-46 with the draw injected
    zs = rearrange(z, '... c (i pi) (j pj) -> ... (c pi pj) i j', pi=2, pj=2)
    recon = ref.decode(ref._normalize_latent(zs), wvs)   # new_autoencoder.py:466-478
    char = Charb()(recon, x)                             # consistency_loss.py:12-21 (eps 1e-3)
    l1 = F.l1_loss(recon, x)
    kl = post.kl()
    loss = char + 1e-6 * kl.mean()
    loss.backward()
    named = [('encoder.' + k, p) for k, p in ref.encoder.named_parameters()] + \
            [('decoder.' + k, p) for k, p in ref.decoder.named_parameters()]
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)))
    names = np.array([k for k, _ in named])
    norms = np.array([p.grad.double().norm().item() for _, p in named])
    samples = np.stack([npy(p.grad).reshape(-1)[sample_index(p.numel())] for _, p in named])
    bn_mean, bn_var = npy(ref.bn.running_mean), npy(ref.bn.running_var)

    # the oracle on the same step (asserted, as for every other fixture)
    mine = O.build_vae(cfg)
    O.det_fill(mine, seed=31)
    mine.train()
    rec_o, post_o = mine(x, wvs, noise=eps)
    loss_o = O.charbonnier(rec_o, x) + 1e-6 * post_o.kl().mean()
    loss_o.backward()
    close(rec_o, recon, 1e-4, 'full_step.recon')
    close(loss_o, loss, 2e-5, 'full_step.loss')
    po = dict(mine.named_parameters())
    tot = gnorm
    worst = 0.0
    for (k, p), n in zip(named, norms):
        err = (po[k].grad.double() - p.grad.double()).norm().item()
        worst = max(worst, err / max(n, 1e-3 * tot))
    assert worst < 2e-4, f'oracle != reference for full_step gradients: {worst}'
    print(f'  oracle vs reference: worst per-tensor gradient error {worst:.2e}')
    save('full_cfg2_step', x_checksum=np.float64(x.double().sum().item()), eps=npy(eps),
         recon_sub=npy(recon[:, :, ::8, ::8]), loss=np.float64(loss.item()), charbonnier=np.float64(char.item()),
         l1=np.float64(l1.item()), kl=npy(kl), grad_norm=np.float64(gnorm), grad_names=names,
         grad_norms=norms, grad_samples=samples, bn_running_mean=bn_mean, bn_running_var=bn_var)


TRAJ_PROBES = [   # ~20 tensors across the model: 3x3 / 1x1 weights at every level, norms, biases, both hyper-networks
    'encoder.conv_in.weight_generator.fc_weight.weight', 'encoder.conv_in.weight_generator.transformer_encoder.layers.0.linear1.weight',
    'encoder.down.0.block.0.conv1.weight', 'encoder.down.0.block.1.conv2.weight', 'encoder.down.0.downsample.conv.weight',
    'encoder.down.1.block.0.nin_shortcut.weight', 'encoder.down.1.block.0.norm2.weight', 'encoder.down.2.block.1.conv1.weight',
    'encoder.down.3.block.1.conv1.weight', 'encoder.mid.attn_1.q.weight', 'encoder.mid.block_2.conv2.bias', 'encoder.conv_out.weight',
    'decoder.conv_in.weight', 'decoder.mid.attn_1.proj_out.weight', 'decoder.up.3.block.0.conv1.weight', 'decoder.up.3.upsample.conv.weight',
    'decoder.up.2.block.2.conv2.weight', 'decoder.up.1.block.0.nin_shortcut.weight', 'decoder.up.1.upsample.conv.weight',
    'decoder.up.0.block.2.conv1.weight', 'decoder.norm_out.weight', 'decoder.conv_out.weight_generator.fc_weight.weight',
]


TRAJ_LR = 1e-5   # see full_traj's docstring
TRAJ_SAMPLES = 256


def full_traj(R):
    """THREE consecutive optimiser steps of the reference at full size: the step sequence of
    new_autoencoder.py:587-652 -- forward with the posterior noise injected, Charbonnier + 1e-6 KL, backward,
    clip_grad_norm_(1.0), Adam.step() (configure_optimizers :549-556) -- at a constant LR of 1e-5: the reference's own
    training ramps the LR from 0 to base_lr = 1e-4 over 2000 warm-up steps (5e-8 per step: three steps from 0 would move
    nothing), and a constant 1e-4 on the closed-form fill is outside any regime the reference trains in (generated and looked
    at: loss 2.2 -> 10.9 -> 3.8, KL x 300, and the fp32 oracle already drifts 35 x per step from the fp32 reference) -- 1e-5 is
    the schedule's value at step 200, moves every weight by ~3e-4 of its RMS per step, and stays stable.  On the
    configs/eo-vae.yaml model, B = 2, S2RGB 256x256, a new batch and a new draw per step.  Stored per step: loss, its
    terms, the pre-clip gradient norm; after step 3: for ~20 named tensors the norm of (p3 - p0) and 256 sampled entries of it,
    the norm of p0, and a subsample of the THIRD step's reconstruction (it saw two updates).  What it pins that the one-step
    fixture cannot: what the gradient arithmetic (one-f16-product weight gradients feeding Adam) does to the trajectory."""
    print('full-size cfg2 three-step trajectory fixture (B=2, S2RGB, 256x256) -- takes ~5 min')
    from einops import rearrange
    F = torch.nn.functional
    cfg = O.model_cfg()
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    O.det_fill(ref, seed=31)
    ref.train()
    ns = {'torch': torch, 'nn': torch.nn, 'F': F}
    Charb = ref_class_from_source(os.path.join(REF, 'eo_vae/models/modules/consistency_loss.py'), 'CharbonnierLoss', ns)()
    mine = O.build_vae(cfg)
    O.det_fill(mine, seed=31)
    mine.train()
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    params_r = [p for p in ref.encoder.parameters() if p.requires_grad] + [p for p in ref.decoder.parameters() if p.requires_grad]
    params_o = [p for p in mine.encoder.parameters() if p.requires_grad] + [p for p in mine.decoder.parameters() if p.requires_grad]
    lr = float(os.environ.get('EOVAE_TRAJ_LR', TRAJ_LR))
    opt_r = torch.optim.Adam(params_r, lr=lr)            # new_autoencoder.py:556
    opt_o = torch.optim.Adam(params_o, lr=lr)
    named_r = dict([('encoder.' + k, p) for k, p in ref.encoder.named_parameters()] +
                   [('decoder.' + k, p) for k, p in ref.decoder.named_parameters()])
    named_o = dict([('encoder.' + k, p) for k, p in mine.encoder.named_parameters()] +
                   [('decoder.' + k, p) for k, p in mine.decoder.named_parameters()])
    assert all(k in named_r for k in TRAJ_PROBES), [k for k in TRAJ_PROBES if k not in named_r]
    p0 = {k: named_r[k].detach().clone() for k in TRAJ_PROBES}
    out = {}
    for step in range(3):
        x = O.synthetic_batch(2, 3, 256, seed=1300 + step)
        eps = hashed((2, 32, 32, 32), 1700 + step, 1.0)
        post = ref.encode(x, wvs)
        z = post.mean + post.std * eps                       # distributions.py:44-46 with the draw injected
        zs = rearrange(z, '... c (i pi) (j pj) -> ... (c pi pj) i j', pi=2, pj=2)
        recon = ref.decode(ref._normalize_latent(zs), wvs)   # new_autoencoder.py:466-478
        opt_r.zero_grad()
        char = Charb(recon, x)
        kl = post.kl().mean()
        loss = char + 1e-6 * kl
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(params_r, 1.0)   # :648-651
        opt_r.step()
        # the oracle on the same step
        rec_o, post_o = mine(x, wvs, noise=eps)
        opt_o.zero_grad()
        loss_o = O.charbonnier(rec_o, x) + 1e-6 * post_o.kl().mean()
        loss_o.backward()
        gn_o = torch.nn.utils.clip_grad_norm_(params_o, 1.0)
        opt_o.step()
        e_l = close(loss_o, loss, 2e-5, f'full_traj.loss[{step}]')
        e_g = close(gn_o, gn, 1e-3, f'full_traj.gnorm[{step}]')
        e_r = close(rec_o, recon, 2e-4, f'full_traj.recon[{step}]')
        print(f'  step {step}: loss {loss.item():.6f} gnorm {gn.item():.4f}; oracle vs reference: loss {e_l:.1e} gnorm {e_g:.1e} recon {e_r:.1e}')
        out[f'x_checksum{step}'] = np.float64(x.double().sum().item())
        out[f'eps{step}'] = npy(eps)
        out[f'loss{step}'] = np.float64(loss.item())
        out[f'charbonnier{step}'] = np.float64(char.item())
        out[f'kl{step}'] = np.float64(kl.item())
        out[f'gnorm{step}'] = np.float64(gn.item())
        if step == 2:
            out['recon2_sub'] = npy(recon[:, :, ::8, ::8])
    worst = 0.0
    dn, d0, ds = [], [], []
    for k in TRAJ_PROBES:
        d_r = (named_r[k].detach() - p0[k]).double()
        d_o = (named_o[k].detach() - p0[k]).double()
        worst = max(worst, (d_o - d_r).norm().item() / d_r.norm().item())
        dn.append(d_r.norm().item())
        d0.append(p0[k].double().norm().item())
        ds.append(npy(d_r.float()).reshape(-1)[sample_index(d_r.numel(), TRAJ_SAMPLES)])
    print(f'  oracle vs reference: worst parameter-delta error after 3 steps {worst:.2e} (relative to the delta\'s norm)')
    assert worst < 0.05, worst
    out.update(lr=np.float64(lr), probe_names=np.array(TRAJ_PROBES), delta_norms=np.array(dn), p0_norms=np.array(d0), delta_samples=np.stack(ds),
               oracle_worst_delta_err=np.float64(worst),
               bn_running_mean=npy(ref.bn.running_mean), bn_running_var=npy(ref.bn.running_var))
    save('full_cfg2_traj', **out)



def full_step12(R):
    """cfg3's arithmetic path at FULL size on the reference itself: the same train step as
    full_step, on S2L2A (12 wavelengths, terramesh_datamodule.py:21-34), B=2, 256x256.  At C = 12 the first / last
    layers are 12-channel dynamic convolutions (new_autoencoder.py:587-690 -> dynamic_conv.py:499-535, 666-710).
    Loss = Charbonnier + 1e-6 KL as in full_step (L1 stored beside it: its sign() gradient flips on round-off,
    DESIGN.md s2).  The MS-SSIM term of cfg3 is NOT part of this fixture: consistency_loss.py:6 imports torchmetrics,
    which is not installed ("parity unpinned")."""
    print('full-size cfg3 train-step fixture (B=2, S2L2A 12 ch, 256x256, forward + backward) -- takes ~2 min')
    from einops import rearrange
    F = torch.nn.functional
    cfg = O.model_cfg()
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    O.det_fill(ref, seed=31)
    ref.train()
    ns = {'torch': torch, 'nn': torch.nn, 'F': F}
    Charb = ref_class_from_source(os.path.join(REF, 'eo_vae/models/modules/consistency_loss.py'),
                                  'CharbonnierLoss', ns)
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
    assert wvs.numel() == 12
    x = O.synthetic_batch(2, 12, 256, seed=1236)
    eps = hashed((2, 32, 32, 32), 1612, 1.0)
    post = ref.encode(x, wvs)
    z = post.mean + post.std * eps
    zs = rearrange(z, '... c (i pi) (j pj) -> ... (c pi pj) i j', pi=2, pj=2)
    recon = ref.decode(ref._normalize_latent(zs), wvs)
    char = Charb()(recon, x)
    l1 = F.l1_loss(recon, x)
    kl = post.kl()
    loss = char + 1e-6 * kl.mean()
    loss.backward()
    named = [('encoder.' + k, p) for k, p in ref.encoder.named_parameters()] + \
            [('decoder.' + k, p) for k, p in ref.decoder.named_parameters()]
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)))
    names = np.array([k for k, _ in named])
    norms = np.array([p.grad.double().norm().item() for _, p in named])
    samples = np.stack([npy(p.grad).reshape(-1)[sample_index(p.numel())] for _, p in named])
    bn_mean, bn_var = npy(ref.bn.running_mean), npy(ref.bn.running_var)
    mine = O.build_vae(cfg)
    O.det_fill(mine, seed=31)
    mine.train()
    rec_o, post_o = mine(x, wvs, noise=eps)
    loss_o = O.charbonnier(rec_o, x) + 1e-6 * post_o.kl().mean()
    loss_o.backward()
    close(rec_o, recon, 1e-4, 'full_step12.recon')
    close(loss_o, loss, 2e-5, 'full_step12.loss')
    po = dict(mine.named_parameters())
    worst = 0.0
    for (k, p), n in zip(named, norms):
        err = (po[k].grad.double() - p.grad.double()).norm().item()
        worst = max(worst, err / max(n, 1e-3 * gnorm))
    assert worst < 2e-4, f'oracle != reference for full_step12 gradients: {worst}'
    print(f'  oracle vs reference: worst per-tensor gradient error {worst:.2e}')
    save('full_cfg3_step', x_checksum=np.float64(x.double().sum().item()), eps=npy(eps),
         recon_sub=npy(recon[:, :, ::8, ::8]), loss=np.float64(loss.item()), charbonnier=np.float64(char.item()),
         l1=np.float64(l1.item()), kl=npy(kl), grad_norm=np.float64(gnorm), grad_names=names,
         grad_norms=norms, grad_samples=samples, bn_running_mean=bn_mean, bn_running_var=bn_var)


def full_encode(R):
    """cfg5's path at FULL size on the reference itself: encode_latents.py:160-169 (`encode_spatial_norm` =
    EOFluxVAE.encode_spatial_normalized, new_autoencoder.py:480-502) on two S2L2A batches of 2 x 12 x 256 x 256,
    then RunningStatsButFast (encode_latents.py:65-94) over the two latent batches."""
    print('full-size cfg5 encode fixture (2 batches of B=2, S2L2A, 256x256) -- takes ~30 s')
    cfg = O.model_cfg()
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    O.det_fill(ref, seed=31)
    ref.eval()
    ns = {'torch': torch, 'nn': torch.nn}
    Ref = ref_class_from_source(os.path.join(REF, 'encode_latents.py'), 'RunningStatsButFast', ns)
    stats, mine_stats = Ref((32,), [0, 2, 3]), O.RunningStats(32)
    mine = O.build_vae(cfg)
    mine.load_state_dict(ref.state_dict())
    mine.eval()
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
    arrays = {}
    for i in range(2):
        x = O.synthetic_batch(2, 12, 256, seed=1240 + i)
        with torch.no_grad():
            z = ref.encode_spatial_normalized(x, wvs)
            close(mine.encode_spatial_normalized(x, wvs), z, 1e-4, f'full_encode.z{i}')
        assert tuple(z.shape) == (2, 32, 32, 32), z.shape
        stats(z)
        mine_stats.update(z)
        arrays[f'x_checksum{i}'] = np.float64(x.double().sum().item())
        arrays[f'z_sub{i}'] = npy(z[:, :, ::2, ::2])
    for k in ['mean', 'var', 'std', 'min', 'max', 'count']:
        close(getattr(mine_stats, k), getattr(stats, k), 1e-6, 'full_encode.stats.' + k)
        arrays['stats_' + k] = npy(getattr(stats, k))
    save('full_cfg5_encode', **arrays)


def extra_losses(R):
    """SAM / gradient-difference / focal-frequency branches of EOConsistencyLoss.
    consistency_loss.py cannot be imported as a whole (top-level torchmetrics import), so its
    classes are executed one by one; SSIMLoss (torchmetrics, parity unpinned) is a placeholder
    that the fixture never calls (msssim_weight = 0)."""
    print('extra loss branches')
    F = torch.nn.functional
    ffl = importlib.import_module('eo_vae.models.modules.ffl')
    src = os.path.join(REF, 'eo_vae/models/modules/consistency_loss.py')
    ns = {'torch': torch, 'nn': torch.nn, 'F': F, 'FFL': ffl.FocalFrequencyLoss,
          'SSIMLoss': type('SSIMLoss', (torch.nn.Module,), {})}
    for cls in ['CharbonnierLoss', 'SAMLoss', 'GradientDifferenceLoss', 'EOConsistencyLoss']:
        ref_class_from_source(src, cls, ns)
    arrays = {}
    target = hashed((2, 4, 32, 32), 900, 1.0) + 0.2
    pred0 = target + 0.3 * hashed((2, 4, 32, 32), 901, 1.0)
    arrays.update(pred=npy(pred0), target=npy(target))

    def both(name, ref_fn, my_fn, tol=2e-6):
        pr = pred0.clone().requires_grad_(True)
        lr_ = ref_fn(pr, target)
        lr_.backward()
        pm = pred0.clone().requires_grad_(True)
        lm = my_fn(pm, target)
        lm.backward()
        close(lm, lr_, tol, name)
        assert (pm.grad - pr.grad).abs().max() <= 2e-6 * pr.grad.abs().max() + 1e-12, name + '.grad'
        arrays[name] = np.float64(lr_.item())
        arrays[name + '.grad'] = npy(pr.grad)

    both('sam', ns['SAMLoss'](), O.sam_loss)
    both('gdl', ns['GradientDifferenceLoss'](), O.gradient_difference_loss)
    for pf, alpha in [(2, 1.0), (1, 1.0), (2, 0.5)]:
        ref_ffl = ffl.FocalFrequencyLoss(loss_weight=1.0, alpha=alpha, patch_factor=pf, ave_spectrum=False,
                                         batch_matrix=True, log_matrix=True)  # consistency_loss.py:388-395
        both(f'ffl_pf{pf}_a{alpha}', ref_ffl,
             lambda a, b: O.focal_frequency_loss(a, b, alpha=alpha, patch_factor=pf), tol=5e-6)
    # the weighting logic incl. the 1000-step FFL warm-up (:437-464), half way through it
    kw = dict(pixel_weight=1.0, rec_loss_type='char', spectral_weight=0.1, spatial_weight=0.5,
              freq_weight=2.0, freq_start_step=6000, spatial_start_step=10, spectral_start_step=7000)
    ref_l, my_l = ns['EOConsistencyLoss'](**kw), O.EOConsistencyLoss(**kw)
    wvs = torch.tensor(O.WAVELENGTHS['S1RTC'] + O.WAVELENGTHS['S1RTC'])
    for step in [5, 6500, 7200]:
        tr, logs_r = ref_l(inputs=target, wvs=wvs, reconstructions=pred0, global_step=step, split='train')
        tm, logs_m = my_l(inputs=target, wvs=wvs, reconstructions=pred0, global_step=step, split='train')
        close(tm, tr, 2e-6, f'consistency.total[{step}]')
        assert set(logs_r) == set(logs_m), (sorted(logs_r), sorted(logs_m))
        arrays[f'total_step{step}'] = np.float64(tr.item())
        arrays[f'logkeys_step{step}'] = np.array(sorted(logs_r))
    save('op_losses_extra', **arrays)


def eqvae(R):
    """EQ-VAE branches of the train step (new_autoencoder.py:447-460, 520-531, 601-627) with the
    random draw injected: (scale, angle) -> recon, area/rot90 target, L1 loss, gradient norms."""
    print('EQ-VAE fixtures (tiny model, train mode)')
    F = torch.nn.functional
    cfg = O.model_cfg(**{**O.TINY, 'channels': 3})
    ref = R.vae.EOFluxVAE(R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder']),
                          torch.nn.Identity(), freeze_body=False)
    mine = O.build_vae(cfg)
    O.det_fill(ref, seed=23)
    mine.load_state_dict(ref.state_dict())
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    x = O.synthetic_batch(2, 3, 64, seed=808)
    arrays = {'x': npy(x)}
    # anisotropic scale with an odd angle makes the reference's own target mis-shaped
    # (interpolate to recon's size, THEN rot90), so that draw is not a valid fixture
    for tag, scale, angle in [('a', 0.75, 1), ('b', (0.5, 0.75), 2), ('c', 0.5, None), ('d', 0.375, 3)]:
        ref.load_state_dict(sd0), mine.load_state_dict(sd0)
        ref.train(), mine.train()
        torch.manual_seed(4242)
        eps = torch.randn(2, 32, 8, 8)      # what posterior.sample() will draw (distributions.py:44-46)
        torch.manual_seed(4242)
        recon, post = ref.forward(x, wvs, scale=scale, angle=angle)
        with torch.no_grad():               # new_autoencoder.py:612-617 / 623-627
            tgt = F.interpolate(x, size=recon.shape[-2:], mode='area')
            if angle is not None:
                tgt = torch.rot90(tgt, k=angle, dims=[-1, -2])
        ref.zero_grad()
        loss = F.l1_loss(recon, tgt)
        loss.backward()
        rec_m, _ = mine(x, wvs, scale=scale, angle=angle, noise=eps)
        tgt_m = mine.eqvae_target(x, rec_m.shape[-2:], angle)
        mine.zero_grad()
        loss_m = F.l1_loss(rec_m, tgt_m)
        loss_m.backward()
        close(rec_m, recon, 1e-4, f'eqvae.recon[{tag}]')
        close(tgt_m, tgt, 1e-6, f'eqvae.target[{tag}]')
        close(loss_m, loss, 2e-5, f'eqvae.loss[{tag}]')
        gr = {k: p.grad.norm().item() for k, p in ref.named_parameters() if p.grad is not None}
        gm = {k: p.grad.norm().item() for k, p in mine.named_parameters() if p.grad is not None}
        assert gr.keys() == gm.keys()
        for k in gr:
            assert abs(gr[k] - gm[k]) <= 2e-3 * max(gr.values()), (k, gr[k], gm[k])
        sc = np.array(scale if isinstance(scale, tuple) else (scale, scale), dtype=np.float64)
        arrays.update({f'{tag}.scale': sc, f'{tag}.angle': np.int64(-1 if angle is None else angle),
                       f'{tag}.eps': npy(eps), f'{tag}.recon': npy(recon), f'{tag}.target': npy(tgt),
                       f'{tag}.loss': np.float64(loss.item()),
                       f'{tag}.bn_mean': npy(ref.bn.running_mean)})
        if tag == 'a':
            arrays['a.grad_names'] = np.array(list(gr.keys()))
            arrays['a.grad_norms'] = np.array(list(gr.values()), dtype=np.float64)
    save('tiny_eqvae', **arrays)


def distill(R):
    """Stage-1 weight distillation: the reference's DistillationModule
    (weight_distill_train.py:146-331, executed class by class: the script's top-level imports need
    hydra / lightning) driven the way Lightning's automatic optimisation drives it."""
    print('distillation fixture (tiny dynamic layers)')
    import dataclasses

    class LightningStandIn(torch.nn.Module):   # the three members the module touches
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device('cpu')

    src = os.path.join(REF, 'weight_distill_train.py')
    ns = {'torch': torch, 'nn': torch.nn, 'F': torch.nn.functional, 'Tensor': torch.Tensor,
          'dataclass': dataclasses.dataclass, 'LightningModule': LightningStandIn}
    DistillConfig = ref_class_from_source(src, 'DistillConfig', ns)
    Module = ref_class_from_source(src, 'DistillationModule', ns)
    cfg = O.model_cfg(**{**O.TINY, 'channels': 3})
    enc_r, dec_r = R.model.Encoder(**cfg['encoder']), R.model.Decoder(**cfg['decoder'])
    holder = torch.nn.ModuleDict({'encoder': enc_r, 'decoder': dec_r})
    O.det_fill(holder, seed=41)
    enc_m, dec_m = O.Encoder(**cfg['encoder']), O.Decoder(**cfg['decoder'])
    enc_m.load_state_dict(enc_r.state_dict()), dec_m.load_state_dict(dec_r.state_dict())
    teacher = {'encoder_weight': hashed((32, 3, 3, 3), 950, 0.15), 'encoder_bias': hashed((32,), 951, 0.05),
               'decoder_weight': hashed((3, 32, 3, 3), 952, 0.15), 'decoder_bias': hashed((3,), 953, 0.05)}
    dcfg = DistillConfig(max_steps=10, lr=1e-3)
    ref = Module(enc_r, dec_r, {k: v.clone() for k, v in teacher.items()}, dcfg)
    conf = ref.configure_optimizers()
    opt, sch = conf['optimizer'], conf['lr_scheduler']['scheduler']
    assert conf['lr_scheduler']['interval'] == 'step'
    mine = O.DistillState(enc_m, dec_m, teacher, rgb_wavelengths=dcfg.rgb_wavelengths, lr=1e-3, max_steps=10)
    arrays = {k: npy(v) for k, v in teacher.items()}
    arrays['rgb_wavelengths'] = np.array(dcfg.rgb_wavelengths, dtype=np.float64)
    for step in range(4):
        loss, logs = ref.compute_distillation_loss()
        opt.zero_grad()
        loss.backward()
        lr = opt.param_groups[0]['lr']
        opt.step()
        sch.step()
        loss_m, logs_m, lr_m = mine.step()
        close(loss_m, loss, 1e-5, f'distill.loss[{step}]')
        assert abs(lr - lr_m) < 1e-12 and set(logs) == set(logs_m)
        for k in logs:
            close(logs_m[k], logs[k], 1e-5, f'distill.{k}[{step}]')
            arrays[f'step{step}.{k}'] = np.float64(logs[k].item())
        arrays[f'step{step}.lr'] = np.float64(lr)
    with torch.no_grad():
        w, b = enc_r.conv_in.get_distillation_weight(ref.rgb_wvs)
        arrays['final.enc_w'], arrays['final.enc_b'] = npy(w), npy(b)
        w, b = dec_r.conv_out.get_distillation_weight(ref.rgb_wvs)
        arrays['final.dec_w'], arrays['final.dec_b'] = npy(w), npy(b)
        wm, bm = enc_m.conv_in.get_distillation_weight(mine.wvs)
        close(wm, arrays['final.enc_w'], 1e-3, 'distill.final.enc_w')   # AdamW: lr-sized steps on round-off
        wm, bm = dec_m.conv_out.get_distillation_weight(mine.wvs)
        close(bm, arrays['final.dec_b'], 1e-3, 'distill.final.dec_b')
    arrays['logkeys'] = np.array(sorted(logs))
    save('distill_steps', **arrays)


def running_stats():
    print('RunningStatsButFast fixture')
    ns = {'torch': torch, 'nn': torch.nn}
    Ref = ref_class_from_source(os.path.join(REF, 'encode_latents.py'), 'RunningStatsButFast', ns)
    ref = Ref((32,), [0, 2, 3])
    mine = O.RunningStats(32)
    arrays = {}
    for i in range(3):
        z = hashed((4, 32, 8, 8), 700 + i, 1.0 + i) + 0.3 * i
        ref(z)
        mine.update(z)
        arrays[f'z{i}'] = npy(z)
    for k in ['mean', 'var', 'std', 'min', 'max', 'count']:
        close(getattr(mine, k), getattr(ref, k), 1e-6, 'stats.' + k)
        arrays[k] = npy(getattr(ref, k))
    save('running_stats', **arrays)


def batch_prep():
    """N2: what the reference's collate does to a raw S2L2A batch -- Sentinel2L2ANorm (clip +
    z-score, terramesh_datamodule.py:130-194), bilinear resize to target_size (:476-479), the D4 augmentation
    (:347-369) -- run from the reference's own source for 8 seeded draws."""
    print('batch preparation fixture')
    import random
    ns = {'torch': torch, 'nn': torch.nn, 'random': random}
    Norm, aug = ref_defs_from_source(os.path.join(REF, 'eo_vae', 'datasets', 'terramesh_datamodule.py'),
                                     ['Sentinel2L2ANorm', 'apply_batch_augmentations'], ns)
    norm = Norm()
    raw = hashed((2, 12, 20, 28), 811, 1.0) * 4000.0 + 3000.0      # spans below 0 and above 10000
    arrays = {'raw': npy(raw), 'mean': npy(norm.mean.reshape(-1)), 'std': npy(norm.std.reshape(-1)),
              'target_size': np.array([24, 32])}
    x = norm(raw)
    x = torch.nn.functional.interpolate(x, size=(24, 32), mode='bilinear', align_corners=False)
    for seed in range(8):
        random.seed(seed)
        out = aug(x)
        random.seed(seed)                       # replay the three draws the function made
        hf, vf, k = random.random() > 0.5, random.random() > 0.5, random.randint(0, 3)
        arrays[f'draw{seed}'] = np.array([int(hf), int(vf), k])
        arrays[f'out{seed}'] = npy(out)
    # round 4: the other normalisers and the random-modality collate itself, from the reference's source
    import numpy
    ns2 = {'torch': torch, 'nn': torch.nn, 'random': random, 'np': numpy}
    names = ['WAVELENGTHS', 'NORM_STATS_LEGACY', 'Sentinel2L2ANorm', 'Sentinel2L1CNorm', 'LegacyZScoreNorm', 'NormalizerFactory',
             'apply_batch_augmentations', 'single_modality_collate_fn']
    ref = dict(zip(names, ref_defs_from_source(os.path.join(REF, 'eo_vae', 'datasets', 'terramesh_datamodule.py'), names, ns2)))
    raw_l1c = hashed((2, 13, 20, 28), 812, 1.0) * 3500.0 + 2500.0
    raw_s1 = hashed((2, 2, 20, 28), 813, 1.0) * 6.0 - 14.0          # dB
    arrays['raw_l1c'], arrays['raw_s1rtc'] = npy(raw_l1c), npy(raw_s1)
    arrays['norm_l1c_custom'] = npy(ref['Sentinel2L1CNorm']()(raw_l1c))
    arrays['norm_s1rtc_legacy'] = npy(ref['LegacyZScoreNorm']('S1RTC')(raw_s1))
    arrays['norm_l2a_legacy'] = npy(ref['NormalizerFactory'].create('S2L2A', 'legacy')(raw))
    assert type(ref['NormalizerFactory'].create('S1RTC', 'custom')).__name__ == 'LegacyZScoreNorm'   # the fall-back (:320-327)
    # single_modality_collate_fn(['S2L2A', 'S1RTC'], custom scheme, target 24 x 32, train) on the raw batch, 8 seeds
    collate = ref['single_modality_collate_fn'](['S2L2A', 'S1RTC'], normalize=True, norm_scheme='custom', target_size=(24, 32),
                                                mode='train')
    for seed in range(8):
        random.seed(100 + seed)
        out = collate({'S2L2A': raw.clone(), 'S1RTC': raw_s1.clone()})
        arrays[f'collate{seed}_image'] = npy(out['image'])
        arrays[f'collate{seed}_wvs'] = npy(out['wvs'])
        arrays[f'collate{seed}_modality'] = np.array(out['modality'])
    # eval mode, one modality, key 'image', legacy scheme, no resize
    collate1 = ref['single_modality_collate_fn'](['S2L1C'], normalize=True, norm_scheme='legacy', target_size=None, mode='eval')
    arrays['collate_eval_l1c'] = npy(collate1({'image': raw_l1c.clone()})['image'])
    save('batch_prep', **arrays)


if __name__ == '__main__':
    torch.set_num_threads(8)
    R = import_reference()
    torch.manual_seed(0)  # dynamic_conv.py:7-8 reseeds the global RNG at import (quirk q1)
    only = sys.argv[1:]  # e.g. `gen_golden.py extra_losses eqvae` regenerates just those groups
    groups = [('per_op', lambda: per_op(R)), ('whole_models', lambda: whole_models(R)),
              ('extra_losses', lambda: extra_losses(R)), ('eqvae', lambda: eqvae(R)),
              ('distill', lambda: distill(R)),
              ('running_stats', running_stats), ('batch_prep', batch_prep),
              ('full_size', lambda: full_size(R)), ('full_step', lambda: full_step(R)),
              ('full_step12', lambda: full_step12(R)), ('full_encode', lambda: full_encode(R)),
              ('full_traj', lambda: full_traj(R))]
    for name, fn in groups:
        if not only or name in only:
            fn()
    print('all fixtures written; oracle == reference on every one of them')
