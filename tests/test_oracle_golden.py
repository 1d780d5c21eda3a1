"""CPU: the oracle restatement (oracle/eo_vae_cpu.py) against the golden vectors that
tests/golden/gen_golden.py produced by running the reference itself."""
import numpy as np
import pytest
import torch

from oracle import eo_vae_cpu as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def assert_close(a, b, rtol, what=''):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1.0)
    assert err <= rtol * scale, f'{what}: max err {err:.3e} > {rtol:.1e} * {scale:.3e}'


def _module_case(golden, name, module, seed, gain=1.0):
    g = golden(name)
    O.det_fill(module, seed=seed, gain=gain)
    x = T(g['x']).requires_grad_(True)
    y = module(x)
    assert_close(y, g['y'], 1e-5, name + '.y')
    y.backward(T(g['gy']))
    assert_close(x.grad, g['gx'], 1e-4, name + '.gx')
    for k, p in module.named_parameters():
        assert_close(p.grad, g['g.' + k], 1e-3, f'{name}.g[{k}]')


@pytest.mark.parametrize('name,cin,cout', [('op_res_same', 32, 32), ('op_res_proj', 32, 64)])
def test_resnet_block(golden, name, cin, cout):
    _module_case(golden, name, O.ResnetBlock(cin, cout), seed=11)


@pytest.mark.parametrize('name', ['op_attn_8', 'op_attn_16'])
def test_attn_block(golden, name):
    _module_case(golden, name, O.AttnBlock(64), seed=12, gain=2.0)


def test_down_up(golden):
    _module_case(golden, 'op_down', O.Downsample(32), seed=13)
    _module_case(golden, 'op_up', O.Upsample(32), seed=13)


def test_dynamic_convs(golden):
    g = golden('op_dynconv')
    din = O.DynamicConv(wv_planes=256, embed_dim=128, num_layers=4).eval()
    dout = O.DynamicConv_decoder(wv_planes=256, embed_dim=128, num_layers=4).eval()
    O.det_fill(din, seed=14)
    O.det_fill(dout, seed=15)
    for mod, wl in O.WAVELENGTHS.items():
        wvs = torch.tensor(wl)
        with torch.no_grad():
            w, b = din.generate(wvs)
            assert_close(w, g[f'{mod}.w_in'], 1e-5, mod + '.w_in')
            assert_close(b, g[f'{mod}.b_in'], 1e-5, mod + '.b_in')
            w, b = dout.generate(wvs)
            assert_close(w, g[f'{mod}.w_out'], 1e-5, mod + '.w_out')
            assert_close(b, g[f'{mod}.b_out'], 1e-5, mod + '.b_out')   # scaled twice (q4)
            assert_close(din(T(g[f'{mod}.x']), wvs), g[f'{mod}.y_in'], 1e-5, mod + '.y_in')
            assert_close(dout(T(g[f'{mod}.h']), wvs), g[f'{mod}.y_out'], 1e-5, mod + '.y_out')


def test_posterior(golden):
    g = golden('op_posterior')
    p = O.DiagonalGaussianDistribution(T(g['moments']))
    assert_close(p.sample(T(g['eps'])), g['sample'], 1e-6, 'sample')
    assert_close(p.kl(), g['kl'], 1e-6, 'kl')
    assert_close(p.mode(), g['mode'], 0.0, 'mode')


@pytest.fixture(scope='module')
def tiny():
    m = O.build_vae(O.model_cfg(**O.TINY))
    O.det_fill(m, seed=21)
    return m


def test_tiny_eval(golden, tiny):
    g = golden('tiny_eval')
    tiny.eval()
    for mod in ['S2RGB', 'S1RTC', 'S2L2A']:
        wvs = torch.tensor(O.WAVELENGTHS[mod])
        x = T(g[f'{mod}.x'])
        # the synthetic generator must regenerate the stored input bit-for-bit
        assert torch.equal(x, O.synthetic_batch(2, len(wvs), 64, seed=1234 + len(wvs)))
        assert_close(tiny.reconstruct(x, wvs), g[f'{mod}.recon'], 1e-4, mod + '.recon')
        assert_close(tiny.encode_to_latent(x, wvs), g[f'{mod}.latent'], 1e-4, mod + '.latent')
        sp = tiny.encode_spatial_normalized(x, wvs)
        assert_close(sp, g[f'{mod}.spatial'], 1e-4, mod + '.spatial')
        assert_close(tiny.encode(x, wvs).parameters, g[f'{mod}.moments'], 1e-4, mod + '.moments')
        assert_close(tiny.decode_spatial_normalized(T(g[f'{mod}.spatial']), wvs), g[f'{mod}.dec'],
                     1e-4, mod + '.dec')


def test_tiny_train_trajectory(golden):
    g = golden('tiny_train')
    m = O.build_vae(O.model_cfg(**O.TINY))
    O.det_fill(m, seed=21)
    m.train()
    st = O.TrainState(m, base_lr=1e-3, final_lr=2e-4, warmup_epochs=1, decay_end_epoch=3,
                      clip_grad=1.0, steps_per_epoch=2)
    loss_fn = O.EOConsistencyLoss(pixel_weight=1.0, rec_loss_type='l1')
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
    for step in range(3):
        if step == 0:
            recon, post = m(T(g['x0']), wvs, noise=T(g['eps0']))
            assert_close(recon, g['recon0'], 1e-4, 'recon0')
            assert_close(post.kl(), g['kl0'], 1e-5, 'kl0')
            loss = torch.nn.functional.l1_loss(recon, T(g['x0']))
            loss.backward()
            named = {('encoder.' + k): p for k, p in m.encoder.named_parameters()}
            named.update({('decoder.' + k): p for k, p in m.decoder.named_parameters()})
            tot = float(np.sqrt((g['grad_norms'] ** 2).sum()))
            clip = min(1.0, 1.0 / (float(g['gnorm0']) + 1e-6))   # fixture norms are post-clip
            for k, ref in zip(g['grad_names'], g['grad_norms']):
                got = named[str(k)].grad.norm().item() * clip
                assert abs(got - ref) <= 1e-3 * max(ref, 1e-3 * tot), (k, got, ref)
            m.zero_grad()
            # undo the running-stat update of this probing forward
            O.det_fill(m.bn, seed=0)
            sd = O.build_vae(O.model_cfg(**O.TINY))
            O.det_fill(sd, seed=21)
            m.bn.load_state_dict(sd.bn.state_dict())
        loss, logs, gn = st.step(loss_fn, T(g[f'x{step}']), wvs, noise=T(g[f'eps{step}']))
        assert abs(loss.item() - g[f'loss{step}']) <= 1e-4 * abs(g[f'loss{step}']), step  # loss scalars: 1e-4
        assert abs(gn.item() - g[f'gnorm{step}']) <= 1e-3 * g[f'gnorm{step}']
        assert abs(logs['train/lr'] - g[f'lr{step}']) < 1e-12
    assert g['lr0'] == 0.0  # quirk q6: LR factor 0 at step 0
    sd = m.state_dict()
    assert_close(sd['bn.running_mean'], g['bn_running_mean'], 1e-4, 'bn.running_mean')
    assert_close(sd['bn.running_var'], g['bn_running_var'], 1e-4, 'bn.running_var')
    assert int(sd['bn.num_batches_tracked']) == int(g['bn_batches'])
    for k in [k for k in g if k.startswith('final.')]:
        assert_close(sd[k[6:]].reshape(-1)[:64], g[k], 2e-3, k)


def test_running_stats(golden):
    g = golden('running_stats')
    rs = O.RunningStats(32)
    for i in range(3):
        rs.update(T(g[f'z{i}']))
    for k in ['mean', 'var', 'std', 'min', 'max', 'count']:
        assert_close(getattr(rs, k), g[k], 1e-6, k)


def test_state_dict_contract(golden):
    """363 keys / 95 514 145 trainable parameters."""
    g = golden('full_cfg1')
    m = O.build_vae(O.model_cfg())
    assert [str(k) for k in g['keys']] == list(m.state_dict().keys())
    n = sum(p.numel() for p in m.encoder.parameters()) + sum(p.numel() for p in m.decoder.parameters())
    assert n == int(g['n_params']) == 95514145


def test_full_cfg1_reconstruct(golden):
    """BASELINE config 1: S2RGB 3-ch 256x256, batch 2, reconstruct() on CPU."""
    g = golden('full_cfg1')
    m = O.build_vae(O.model_cfg()).eval()
    O.det_fill(m, seed=31)
    torch.set_num_threads(8)
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    x = O.synthetic_batch(2, 3, 256, seed=1234)
    assert abs(x.double().sum().item() - float(g['x_checksum'])) < 1e-6
    rec = m.reconstruct(x, wvs)
    assert_close(rec[:, :, ::8, ::8], g['recon_sub'], 1e-4, 'recon')
    assert_close(m.encode_spatial_normalized(x, wvs)[:, :, ::2, ::2], g['spatial_sub'], 1e-4, 'spatial')


def test_full_cfg2_step_fixture(golden):
    """BASELINE config 2's step at full size (B=2): the oracle's forward + backward against the fixture the
    REFERENCE wrote (loss scalars, reconstruction, every parameter gradient's norm) -- ~45 s on 8 threads."""
    g = golden('full_cfg2_step')
    assert abs(float(np.sqrt((g['grad_norms'] ** 2).sum())) - float(g['grad_norm'])) < 1e-9 * float(g['grad_norm'])
    torch.set_num_threads(8)
    m = O.build_vae(O.model_cfg()).train()
    O.det_fill(m, seed=31)
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB'])
    x = O.synthetic_batch(2, 3, 256, seed=1234)
    rec, post = m(x, wvs, noise=torch.from_numpy(g['eps']))
    char = O.charbonnier(rec, x)
    loss = char + 1e-6 * post.kl().mean()
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 2e-5 * abs(float(g['loss']))
    assert abs(torch.nn.functional.l1_loss(rec, x).item() - float(g['l1'])) <= 2e-5 * float(g['l1'])
    assert_close(rec[:, :, ::8, ::8], g['recon_sub'], 1e-4, 'recon')
    po = dict(m.named_parameters())
    tot = float(g['grad_norm'])
    for k, n in zip(g['grad_names'], g['grad_norms']):
        got = po[str(k)].grad.double().norm().item()
        assert abs(got - n) <= 2e-4 * max(n, 1e-3 * tot), (k, got, n)


def test_full_cfg2_traj_fixture(golden):
    """The three-step trajectory fixture (gen_golden.py full_traj: the REFERENCE's step sequence at full size, LR 1e-5).  The
    generator asserted oracle == reference while writing it (loss 1e-6, update 1.9e-4); replaying three full-size steps
    takes minutes on 8 threads, so the CPU suite checks what is cheap: the three batches are the ones the oracle's
    synthetic_batch regenerates, the probe tensors exist in the oracle's model, and the stored trajectory is self-consistent
    (loss = Charbonnier + 1e-6 KL per step; every update moved, none exploded)."""
    g = golden('full_cfg2_traj')
    for step in range(3):
        x = O.synthetic_batch(2, 3, 256, seed=1300 + step)
        assert abs(x.double().sum().item() - float(g[f'x_checksum{step}'])) < 1e-6
        assert abs(float(g[f'charbonnier{step}']) + 1e-6 * float(g[f'kl{step}']) - float(g[f'loss{step}'])) < 1e-6 * float(g[f'loss{step}'])
        assert g[f'eps{step}'].shape == (2, 32, 32, 32) and 10.0 < float(g[f'gnorm{step}']) < 1e3
    names = dict(O.build_vae(O.model_cfg()).named_parameters())
    assert len(g['probe_names']) == 22 and g['delta_samples'].shape == (22, 256)
    assert all(str(k) in names for k in g['probe_names'])
    lr = float(g['lr'])
    assert lr == 1e-5 and float(g['oracle_worst_delta_err']) < 1e-3
    # Adam moves every entry by at most ~lr per step: 0 < ||p3 - p0|| <= 3 lr sqrt(numel), and far below ||p0||
    assert (g['delta_norms'] > 0).all() and (g['delta_norms'] < 0.05 * g['p0_norms']).all()
    assert np.abs(g['delta_samples']).max() <= 3.0 * lr * 1.01


def test_full_cfg3_step_fixture(golden):
    """BASELINE config 3's arithmetic path at full size (B=2, S2L2A: 12-channel dynamic convolutions first and last):
    the oracle's forward + backward against the fixture the REFERENCE wrote (gen_golden.py full_step12).  The MS-SSIM
    term is not in the fixture (torchmetrics absent: parity unpinned) -- ~60 s on 8 threads."""
    g = golden('full_cfg3_step')
    torch.set_num_threads(8)
    m = O.build_vae(O.model_cfg()).train()
    O.det_fill(m, seed=31)
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
    x = O.synthetic_batch(2, 12, 256, seed=1236)
    assert abs(x.double().sum().item() - float(g['x_checksum'])) < 1e-6 * abs(float(g['x_checksum'])) + 1e-6
    rec, post = m(x, wvs, noise=torch.from_numpy(g['eps']))
    loss = O.charbonnier(rec, x) + 1e-6 * post.kl().mean()
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 2e-5 * abs(float(g['loss']))
    assert_close(rec[:, :, ::8, ::8], g['recon_sub'], 1e-4, 'recon')
    po = dict(m.named_parameters())
    tot = float(g['grad_norm'])
    for k, n in zip(g['grad_names'], g['grad_norms']):
        got = po[str(k)].grad.double().norm().item()
        assert abs(got - n) <= 2e-4 * max(n, 1e-3 * tot), (k, got, n)


def test_full_cfg5_encode_fixture(golden):
    """BASELINE config 5's path at full size: encode_spatial_normalized on two S2L2A batches + RunningStatsButFast,
    oracle against the fixture the REFERENCE wrote (gen_golden.py full_encode)."""
    g = golden('full_cfg5_encode')
    torch.set_num_threads(8)
    m = O.build_vae(O.model_cfg()).eval()
    O.det_fill(m, seed=31)
    wvs = torch.tensor(O.WAVELENGTHS['S2L2A'])
    rs = O.RunningStats(32)
    for i in range(2):
        x = O.synthetic_batch(2, 12, 256, seed=1240 + i)
        with torch.no_grad():
            z = m.encode_spatial_normalized(x, wvs)
        assert_close(z[:, :, ::2, ::2], g[f'z_sub{i}'], 1e-4, f'z{i}')
        rs.update(z)
    for k in ['mean', 'var', 'std', 'min', 'max', 'count']:
        assert_close(getattr(rs, k), g['stats_' + k], 1e-5, k)


# ---- MS-SSIM: PARITY UNPINNED (torchmetrics absent) -> known-answer properties ----

def test_msssim_properties():
    x = O.synthetic_batch(2, 4, 192, seed=5)
    y = x + 0.1 * torch.randn(x.shape, generator=torch.Generator().manual_seed(6))
    assert abs(O.msssim(x, x).item() - 1.0) < 1e-6            # identical inputs -> 1
    v = O.msssim(y, x).item()
    assert 0.0 < v < 1.0
    assert abs(O.msssim(x, y).item() - v) < 1e-6              # symmetric
    perm = torch.tensor([1, 0])
    assert abs(O.msssim(y[perm], x[perm]).item() - v) < 1e-6  # batch-order invariant
    taps = O.gaussian_taps()
    assert taps.numel() == 11 and abs(taps.sum().item() - 1) < 1e-6   # kernel_size=5 is ignored (q5)
    # one scale, constant images: mu = value, sigma = 0 -> ssim = (2ab+c1)/(a^2+b^2+c1), cs = 1
    a, b = torch.full((1, 1, 32, 32), 2.0), torch.full((1, 1, 32, 32), 3.0)
    c1, c2 = (0.01 * 6) ** 2, (0.03 * 6) ** 2
    ssim, cs = O._ssim_cs(a, b, taps, c1, c2)
    # fp32 cancellation in E[x^2]-mu^2 against c2=0.0324 limits this to ~2e-4
    assert abs(ssim.item() - (12 + c1) / (13 + c1)) < 5e-4 and abs(cs.item() - 1) < 5e-4
    # uncorrelated images saturate the loss at 1 with zero gradient (SURVEY 8d cfg3 note)
    z = torch.randn(1, 2, 192, 192, generator=torch.Generator().manual_seed(7)).requires_grad_(True)
    w = torch.randn(1, 2, 192, 192, generator=torch.Generator().manual_seed(8))
    val = O.msssim(z, -z.detach())
    assert val.item() == 0.0


FFL_CASES = [(2, 1.0), (1, 1.0), (2, 0.5)]


def test_extra_loss_branches(golden):
    """SAM / gradient-difference / focal-frequency losses + the branch weighting."""
    g = golden('op_losses_extra')
    target = T(g['target'])
    cases = [('sam', O.sam_loss), ('gdl', O.gradient_difference_loss)]
    cases += [(f'ffl_pf{pf}_a{a}', lambda p, t, pf=pf, a=a: O.focal_frequency_loss(p, t, alpha=a, patch_factor=pf))
              for pf, a in FFL_CASES]
    for name, fn in cases:
        pred = T(g['pred']).requires_grad_(True)
        loss = fn(pred, target)
        loss.backward()
        assert_close(loss, g[name], 1e-5, name)
        assert (pred.grad - T(g[name + '.grad'])).abs().max() <= 1e-5 * np.abs(g[name + '.grad']).max(), name
    kw = dict(pixel_weight=1.0, rec_loss_type='char', spectral_weight=0.1, spatial_weight=0.5,
              freq_weight=2.0, freq_start_step=6000, spatial_start_step=10, spectral_start_step=7000)
    loss_fn = O.EOConsistencyLoss(**kw)
    for step in [5, 6500, 7200]:
        total, logs = loss_fn(inputs=target, wvs=None, reconstructions=T(g['pred']), global_step=step)
        assert_close(total, g[f'total_step{step}'], 1e-5, f'total[{step}]')
        assert sorted(logs) == list(g[f'logkeys_step{step}'])


def test_eqvae_branches(golden):
    """EQ-VAE draws of the train step: latent bilinear scale + rot90, area/rot90 target."""
    g = golden('tiny_eqvae')
    vae = O.build_vae(O.model_cfg(**{**O.TINY, 'channels': 3}))
    O.det_fill(vae, seed=23)
    sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
    x, wvs = T(g['x']), torch.tensor(O.WAVELENGTHS['S2RGB'])
    for tag in 'abcd':
        vae.load_state_dict(sd0)
        vae.train()
        sc = tuple(float(v) for v in g[f'{tag}.scale'])
        scale = sc if sc[0] != sc[1] else sc[0]
        angle = None if int(g[f'{tag}.angle']) < 0 else int(g[f'{tag}.angle'])
        recon, _ = vae(x, wvs, scale=scale, angle=angle, noise=T(g[f'{tag}.eps']))
        target = vae.eqvae_target(x, recon.shape[-2:], angle)
        assert recon.shape == T(g[f'{tag}.recon']).shape
        assert_close(recon, g[f'{tag}.recon'], 1e-4, tag + '.recon')
        assert_close(target, g[f'{tag}.target'], 1e-6, tag + '.target')
        loss = torch.nn.functional.l1_loss(recon, target)
        assert_close(loss, g[f'{tag}.loss'], 2e-5, tag + '.loss')
        assert_close(vae.bn.running_mean, g[f'{tag}.bn_mean'], 1e-5, tag + '.bn_mean')
        if tag == 'a':
            vae.zero_grad()
            loss.backward()
            gn = dict(zip(g['a.grad_names'], g['a.grad_norms']))
            top = max(gn.values())
            for k, p in vae.named_parameters():
                if p.grad is not None:
                    assert abs(p.grad.norm().item() - gn[k]) <= 2e-3 * top, k


def _msssim_scipy(pred, target, data_range=6.0, betas=O.MSSSIM_BETAS, sigma=1.5, clamp=True):
    """Second, independent derivation of the torchmetrics MS-SSIM semantics in float64 on
    scipy.ndimage: separable `correlate1d(mode='mirror')` (= reflect padding without the edge sample, what
    F.pad(mode='reflect') does) over the whole image, then the 5-pixel crop.  Not a reference pin -- a check
    that the oracle's conv2d / padding / cropping / pooling bookkeeping says what the description says."""
    from scipy.ndimage import correlate1d
    k = 2 * int(3.5 * sigma + 0.5) + 1
    assert k == 11                                   # kernel_size=5 of consistency_loss.py:32 is ignored
    d = np.arange(k, dtype=np.float64) - (k - 1) / 2
    taps = np.exp(-0.5 * (d / sigma) ** 2)
    taps /= taps.sum()
    pad = (k - 1) // 2
    blur = lambda a: correlate1d(correlate1d(a, taps, axis=-1, mode='mirror'), taps, axis=-2, mode='mirror')  # noqa: E731
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    x, y = pred.double().numpy(), target.double().numpy()
    per_scale = []
    for s in range(len(betas)):
        mx, my = blur(x), blur(y)
        sxx, syy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
        if clamp:
            sxx, syy = np.maximum(sxx, 0.0), np.maximum(syy, 0.0)
        cs_map = (2 * sxy + c2) / (sxx + syy + c2)
        ssim_map = (2 * mx * my + c1) / (mx * mx + my * my + c1) * cs_map
        crop = (Ellipsis, slice(pad, -pad), slice(pad, -pad))
        cs = cs_map[crop].reshape(x.shape[0], -1).mean(-1)
        ssim = ssim_map[crop].reshape(x.shape[0], -1).mean(-1)
        per_scale.append((ssim, cs))
        b, c, h, w = x.shape
        x = x.reshape(b, c, h // 2, 2, w // 2, 2).mean((3, 5))      # avg_pool2d(2)
        y = y.reshape(b, c, h // 2, 2, w // 2, 2).mean((3, 5))
    vals = [np.maximum(cs, 0.0) for _, cs in per_scale[:-1]] + [np.maximum(per_scale[-1][0], 0.0)]
    out = np.prod(np.stack(vals) ** np.asarray(betas)[:, None], axis=0).mean()
    return out, per_scale


@pytest.mark.parametrize('clamp', [True, False])
def test_msssim_against_independent_scipy_derivation(clamp):
    """VERDICT r1 item 8 / ADVICE: cross-check oracle.msssim per scale (cs, ssim) and in total against a second
    from-scratch float64 implementation, for both settings of the version-sensitive variance clamp."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    x = O.synthetic_batch(2, 3, 192, seed=5)
    y = x + 0.3 * torch.randn(x.shape, generator=g) * (1.0 + x.abs())       # correlated but clearly different
    want, per_scale = _msssim_scipy(y, x, clamp=clamp)
    got = O.msssim(y.double(), x.double(), clamp_variance=clamp).item()
    assert 0.05 < want < 0.98                                               # a non-degenerate case
    assert abs(got - want) < 1e-6 * max(1.0, abs(want))      # (the oracle builds its taps in float32: ~1e-8)
    taps = O.gaussian_taps(1.5).double()
    assert taps.numel() == 11
    xs, ys = y.double(), x.double()
    for s, (ssim_w, cs_w) in enumerate(per_scale):
        ssim_o, cs_o = O._ssim_cs(xs, ys, taps, (0.01 * 6.0) ** 2, (0.03 * 6.0) ** 2, clamp)
        assert np.allclose(ssim_o.numpy(), ssim_w, rtol=1e-6, atol=1e-9), s
        assert np.allclose(cs_o.numpy(), cs_w, rtol=1e-6, atol=1e-9), s
        xs, ys = F.avg_pool2d(xs, 2), F.avg_pool2d(ys, 2)
    # flat patches make the variance estimate round to tiny negatives: there the two settings may differ
    flat = torch.full((1, 2, 176, 176), 0.25, dtype=torch.float64)
    assert abs(O.msssim(flat, flat.clone(), clamp_variance=clamp).item() - 1.0) < 1e-6
