"""GPU: Stage-1 weight distillation on the HIP path (mse/l1 loss kernels, fused AdamW, the fit loop's
automatic optimisation) against the reference-generated fixture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def _module(g, max_steps=10, lr=1e-3):
    from oracle import eo_vae_cpu as O
    from eo_vae_amd.distill import DistillConfig, DistillationModule
    from eo_vae_amd.model import Decoder, Encoder
    cfg = O.model_cfg(**{**O.TINY, 'channels': 3})
    enc_o, dec_o = O.Encoder(**cfg['encoder']), O.Decoder(**cfg['decoder'])
    O.det_fill(torch.nn.ModuleDict({'encoder': enc_o, 'decoder': dec_o}), seed=41)
    enc, dec = Encoder(**cfg['encoder']), Decoder(**cfg['decoder'])
    enc.load_state_dict(enc_o.state_dict()), dec.load_state_dict(dec_o.state_dict())
    teacher = {k: T(g[k]) for k in ['encoder_weight', 'encoder_bias', 'decoder_weight', 'decoder_bias']}
    return DistillationModule(enc, dec, teacher, DistillConfig(max_steps=max_steps, lr=lr))


def test_mse_kernel_and_fused_adamw_vs_torch():
    from eo_vae_amd import ops
    from eo_vae_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(3)
    a0, b = torch.randn(5, 7, 3, generator=gen), torch.randn(5, 7, 3, generator=gen)
    ar = a0.clone().requires_grad_(True)
    torch.nn.functional.mse_loss(ar, b).backward()
    ag = a0.cuda().requires_grad_(True)
    loss = ops.mse_loss(ag, b.cuda())
    loss.backward()
    assert abs(loss.item() - torch.nn.functional.mse_loss(a0, b).item()) < 1e-6
    assert (ag.grad.cpu() - ar.grad).abs().max() < 1e-7
    # decoupled weight decay: 5 steps of torch.optim.AdamW vs the fused kernel on the same gradients
    p_ref = [torch.nn.Parameter(torch.randn(33, generator=gen)), torch.nn.Parameter(torch.randn(4, 5, generator=gen))]
    p_gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in p_ref]
    o_ref = torch.optim.AdamW(p_ref, lr=1e-2, weight_decay=0.1)
    o_gpu = FusedAdam(p_gpu, lr=1e-2, weight_decay=0.1)
    for _ in range(5):
        for pr, pg in zip(p_ref, p_gpu):
            gr = torch.randn(pr.shape, generator=gen)
            pr.grad = gr.clone()
            pg.grad.copy_(gr.cuda()) if pg.grad is not None else setattr(pg, 'grad', gr.cuda())
        o_ref.step(), o_gpu.step()
    for pr, pg in zip(p_ref, p_gpu):
        assert (pg.detach().cpu() - pr.detach()).abs().max() < 2e-6


def test_distillation_steps_against_golden(golden):
    g = golden('distill_steps')
    m = _module(g).cuda()
    conf = m.configure_optimizers()
    opt, sch = conf['optimizer'], conf['lr_scheduler']['scheduler']
    for step in range(4):
        loss, logs = m.compute_distillation_loss()
        opt.zero_grad()
        loss.backward()
        lr = opt.param_groups[0]['lr']
        opt.step()
        sch.step()
        assert sorted(logs) == list(g['logkeys'])
        assert abs(lr - float(g[f'step{step}.lr'])) < 1e-12
        for k, v in logs.items():
            ref = float(g[f'step{step}.{k}'])
            assert abs(float(v.detach()) - ref) <= 1e-4 * max(abs(ref), 1e-6), (step, k)
    with torch.no_grad():
        w, _ = m.encoder.conv_in.get_distillation_weight(m.rgb_wvs)
        _, b = m.decoder.conv_out.get_distillation_weight(m.rgb_wvs)
    assert (w.cpu() - T(g['final.enc_w'])).abs().max() < 2e-3 * np.abs(g['final.enc_w']).max()
    assert (b.cpu() - T(g['final.dec_b'])).abs().max() < 2e-3 * np.abs(g['final.dec_b']).max()


def test_distillation_fit_loop_and_checkpoint(tmp_path, golden):
    """Trainer.fit with automatic optimisation reproduces the same 4 steps; the saved `.pt` loads into
    EOFluxVAE and the distilled layers then generate weights closer to the teacher than at start."""
    from eo_vae_amd.autoencoder import EOFluxVAE
    from eo_vae_amd.distill import DummyDataModule, save_distillation_checkpoint
    from eo_vae_amd.model import Decoder, Encoder
    from eo_vae_amd.trainer import Trainer
    from oracle import eo_vae_cpu as O
    g = golden('distill_steps')
    m = _module(g)
    data = DummyDataModule(train_size=4, val_size=2)
    tr = Trainer(max_steps=4, max_epochs=1, default_root_dir=str(tmp_path), log_every_n_steps=1,
                 val_check_interval=2, limit_val_batches=1, enable_checkpointing=False)
    tr.fit(m, train_dataloaders=data.train_dataloader(), val_dataloaders=data.val_dataloader())
    rows = [r for r in tr.history if 'train/total_loss' in r]
    assert [r['step'] for r in rows] == [1, 2, 3, 4]
    for i, r in enumerate(rows):
        ref = float(g[f'step{i}.total_loss'])
        assert abs(r['train/total_loss'] - ref) <= 1e-4 * ref
    assert rows[-1]['train/total_loss'] < rows[0]['train/total_loss']
    assert m.best_loss < float(g['step0.total_loss'])   # validation ran (val_check_interval) and tracked it
    path = str(tmp_path / 'distilled_final.pt')
    save_distillation_checkpoint(m, path, {'model': None})
    cfg = O.model_cfg(**{**O.TINY, 'channels': 3})
    vae = EOFluxVAE(Encoder(**cfg['encoder']), Decoder(**cfg['decoder']), torch.nn.Identity(), ckpt_path=path,
                    freeze_body=False)
    for k, v in m.encoder.conv_in.state_dict().items():
        assert torch.equal(vae.encoder.conv_in.state_dict()[k], v.cpu())
