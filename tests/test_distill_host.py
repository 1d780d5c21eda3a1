"""CPU: Stage-1 weight distillation -- the oracle against the reference-generated
fixture, the teacher-weight loader and the distilled-checkpoint format."""
import numpy as np
import pytest
import torch

from oracle import eo_vae_cpu as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def teacher_of(g):
    return {k: T(g[k]) for k in ['encoder_weight', 'encoder_bias', 'decoder_weight', 'decoder_bias']}


def tiny_io_pair(seed=41):
    cfg = O.model_cfg(**{**O.TINY, 'channels': 3})
    enc, dec = O.Encoder(**cfg['encoder']), O.Decoder(**cfg['decoder'])
    O.det_fill(torch.nn.ModuleDict({'encoder': enc, 'decoder': dec}), seed=seed)
    return cfg, enc, dec


def test_oracle_distillation_steps_against_golden(golden):
    g = golden('distill_steps')
    _, enc, dec = tiny_io_pair()
    st = O.DistillState(enc, dec, teacher_of(g), rgb_wavelengths=tuple(g['rgb_wavelengths']), lr=1e-3, max_steps=10)
    for step in range(4):
        loss, logs, lr = st.step()
        assert sorted(logs) == list(g['logkeys'])
        assert abs(lr - float(g[f'step{step}.lr'])) < 1e-12
        for k, v in logs.items():
            ref = float(g[f'step{step}.{k}'])
            assert abs(float(v) - ref) <= 1e-5 * max(abs(ref), 1e-6), (step, k)
    with torch.no_grad():
        w, _ = enc.conv_in.get_distillation_weight(st.wvs)
        _, b = dec.conv_out.get_distillation_weight(st.wvs)
    assert (w - T(g['final.enc_w'])).abs().max() < 1e-3 * np.abs(g['final.enc_w']).max()
    assert (b - T(g['final.dec_b'])).abs().max() < 1e-3 * np.abs(g['final.dec_b']).max()


def test_load_teacher_weights_formats(tmp_path):
    from safetensors.torch import save_file
    from eo_vae_amd.distill import load_teacher_weights
    sd = {'encoder.conv_in.weight': torch.randn(8, 3, 3, 3), 'encoder.conv_in.bias': torch.randn(8),
          'decoder.conv_out.weight': torch.randn(3, 8, 3, 3), 'decoder.conv_out.bias': torch.randn(3),
          'encoder.mid.block_1.conv1.weight': torch.randn(4, 4, 3, 3)}
    save_file(sd, str(tmp_path / 'ae.safetensors'))
    out = load_teacher_weights(str(tmp_path / 'ae.safetensors'))
    assert torch.equal(out['encoder_weight'], sd['encoder.conv_in.weight'])
    assert torch.equal(out['decoder_bias'], sd['decoder.conv_out.bias'])
    # torch checkpoint under 'state_dict', un-prefixed keys, no biases
    torch.save({'state_dict': {'conv_in.weight': sd['encoder.conv_in.weight'],
                               'conv_out.weight': sd['decoder.conv_out.weight']}}, tmp_path / 't.ckpt')
    out = load_teacher_weights(str(tmp_path / 't.ckpt'))
    assert out['encoder_bias'] is None and out['decoder_bias'] is None
    assert torch.equal(out['decoder_weight'], sd['decoder.conv_out.weight'])
    torch.save({'encoder.conv_in.weight': sd['encoder.conv_in.weight']}, tmp_path / 'half.pt')
    with pytest.raises(ValueError, match='decoder conv_out'):
        load_teacher_weights(str(tmp_path / 'half.pt'))


def test_distilled_checkpoint_format_roundtrip(tmp_path, golden):
    """save_distillation_checkpoint -> the `.pt` layout both load_distilled_checkpoint and
    EOFluxVAE(ckpt_path=...) recognise (new_autoencoder.py:295-356)."""
    from eo_vae_amd.autoencoder import EOFluxVAE
    from eo_vae_amd.distill import (DistillConfig, DistillationModule, load_distilled_checkpoint,
                                    save_distillation_checkpoint)
    from eo_vae_amd.model import Decoder, Encoder
    g = golden('distill_steps')
    cfg, enc_o, dec_o = tiny_io_pair()
    enc, dec = Encoder(**cfg['encoder']), Decoder(**cfg['decoder'])
    enc.load_state_dict(enc_o.state_dict()), dec.load_state_dict(dec_o.state_dict())
    module = DistillationModule(enc, dec, teacher_of(g), DistillConfig(max_steps=10, lr=1e-3))
    module.best_loss = 0.125
    path = str(tmp_path / 'distilled_final.pt')
    save_distillation_checkpoint(module, path, {'model': {'note': 'cfg'}})
    ckpt = torch.load(path, map_location='cpu')
    assert set(ckpt) == {'encoder_conv_in_state_dict', 'decoder_conv_out_state_dict', 'encoder_state_dict',
                         'decoder_state_dict', 'distill_config', 'teacher_shapes', 'model_config'}
    assert ckpt['teacher_shapes'] == {'encoder_weight': (32, 3, 3, 3), 'decoder_weight': (3, 32, 3, 3)}
    assert ckpt['distill_config']['final_loss'] == 0.125
    enc2, dec2 = Encoder(**cfg['encoder']), Decoder(**cfg['decoder'])
    meta = load_distilled_checkpoint(enc2, dec2, path)
    assert meta['distill_config']['max_steps'] == 10
    for k, v in enc.conv_in.state_dict().items():
        assert torch.equal(enc2.conv_in.state_dict()[k], v)
    vae = EOFluxVAE(Encoder(**cfg['encoder']), Decoder(**cfg['decoder']), torch.nn.Identity(), ckpt_path=path,
                    freeze_body=False)
    for k, v in dec.conv_out.state_dict().items():
        assert torch.equal(vae.decoder.conv_out.state_dict()[k], v)
    with pytest.raises(ValueError):  # nothing to optimise
        class NoDyn(torch.nn.Module):
            use_dynamic_ops = False
        DistillationModule(NoDyn(), NoDyn(), teacher_of(g), DistillConfig()).configure_optimizers()
