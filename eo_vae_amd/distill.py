"""Stage-1 weight distillation on the MI355X path (reference
weight_distill_train.py:43-468): a data-free fit of the two dynamic layers' hyper-networks to a
teacher's RGB `conv_in` / `conv_out` kernels, the step that precedes multi-modal fine-tuning.

Same names, arguments and checkpoint layout as the reference script.  The hyper-network itself
runs on PyTorch-ROCm ops; the losses (`ops.mse_loss`, `ops.l1_loss`) and the AdamW
update (`FusedAdam(weight_decay=...)`, one launch over the flat parameter buffer) are HIP kernels."""
from __future__ import annotations

import os
from dataclasses import dataclass
from datetime import datetime

import torch
import torch.nn as nn

from . import ops
from .config import instantiate
from .lightning_compat import LightningModule
from .optim import FusedAdam


@dataclass
class DistillConfig:  # weight_distill_train.py:43-63
    max_steps: int = 5000
    lr: float = 1e-4
    val_every_n_steps: int = 500
    log_every_n_steps: int = 50
    patience: int = 10
    min_delta: float = 1e-7
    rgb_wavelengths: tuple = (0.665, 0.560, 0.490)  # R, G, B (microns), the teacher's channel order
    weight_loss_scale: float = 1.0
    bias_loss_scale: float = 1.0


def load_teacher_weights(ckpt_path: str) -> dict:
    """conv_in / conv_out weight and bias of a pretrained Flux VAE (.safetensors) or a torch
    checkpoint (optionally under 'state_dict'); ValueError when either weight is absent (:71-140)."""
    print(f'Loading teacher weights from: {ckpt_path}')
    if ckpt_path.endswith('.safetensors'):
        from safetensors import safe_open
        sd = {}
        with safe_open(ckpt_path, framework='pt', device='cpu') as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
    else:
        sd = torch.load(ckpt_path, map_location='cpu', weights_only=False)
        sd = sd.get('state_dict', sd)

    def first(*names):
        return next((sd[n] for n in names if n in sd), None)

    out = {'encoder_weight': first('encoder.conv_in.weight', 'conv_in.weight'),
           'encoder_bias': first('encoder.conv_in.bias', 'conv_in.bias'),
           'decoder_weight': first('decoder.conv_out.weight', 'conv_out.weight'),
           'decoder_bias': first('decoder.conv_out.bias', 'conv_out.bias')}
    for side, layer in (('encoder', 'conv_in'), ('decoder', 'conv_out')):
        if out[f'{side}_weight'] is None:
            raise ValueError(f'Could not find {side} {layer} weights in checkpoint.\n'
                             f"Available keys containing 'conv': {[k for k in sd if 'conv' in k.lower()]}")
    for k, v in out.items():
        print(f'  {k} shape: {tuple(v.shape) if v is not None else "None"}')
    return out


class DistillationModule(LightningModule):
    """(:146-331) compares the weights the dynamic layers generate for the RGB wavelengths with the
    teacher's and optimises the dynamic layers only.  Automatic optimisation: the fit loop does
    zero_grad / backward / step / scheduler step."""

    def __init__(self, encoder: nn.Module, decoder: nn.Module, teacher_weights: dict, config: DistillConfig):
        super().__init__()
        self.encoder, self.decoder, self.config = encoder, decoder, config
        self.register_buffer('teacher_enc_w', teacher_weights['encoder_weight'].float())
        self.register_buffer('teacher_dec_w', teacher_weights['decoder_weight'].float())
        for name, key in (('teacher_enc_b', 'encoder_bias'), ('teacher_dec_b', 'decoder_bias')):
            if teacher_weights[key] is not None:
                self.register_buffer(name, teacher_weights[key].float())
            else:
                setattr(self, name, None)
        self.register_buffer('rgb_wvs', torch.tensor(config.rgb_wavelengths, dtype=torch.float32))
        self.best_loss = float('inf')

    def _side(self, tag, layer, teacher_w, teacher_b, logs, total):
        if not hasattr(layer, 'get_distillation_weight'):
            raise AttributeError(f'{tag} dynamic layer ({type(layer).__name__}) does not have '
                                 "'get_distillation_weight' method. This method is required for distillation.")
        w, b = layer.get_distillation_weight(self.rgb_wvs)
        w = w.contiguous()
        loss_w = ops.mse_loss(w, teacher_w)
        total = total + loss_w * self.config.weight_loss_scale
        logs[f'{tag}_weight_loss'] = loss_w
        if b is not None and teacher_b is not None:
            loss_b = ops.mse_loss(b.contiguous().reshape(teacher_b.shape), teacher_b)
            total = total + loss_b * self.config.bias_loss_scale
            logs[f'{tag}_bias_loss'] = loss_b
        with torch.no_grad():
            logs[f'{tag}_weight_mae'] = ops.l1_loss(w.detach(), teacher_w)
            logs[f'{tag}_weight_max_err'] = (w.detach() - teacher_w).abs().max()
        return total

    def compute_distillation_loss(self):  # :190-264
        logs = {}
        total = torch.zeros((), dtype=torch.float32, device=self.rgb_wvs.device)
        if self.encoder.use_dynamic_ops:
            total = self._side('enc', self.encoder.conv_in, self.teacher_enc_w, self.teacher_enc_b, logs, total)
        if self.decoder.use_dynamic_ops:
            total = self._side('dec', self.decoder.conv_out, self.teacher_dec_w, self.teacher_dec_b, logs, total)
        logs['total_loss'] = total
        return total, logs

    def training_step(self, batch, batch_idx):  # the batch is a dummy
        loss, logs = self.compute_distillation_loss()
        self.log_dict({f'train/{k}': v.detach() for k, v in logs.items()}, prog_bar=True, logger=True,
                      on_step=True, on_epoch=False)
        return loss

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            loss, logs = self.compute_distillation_loss()
        if loss < self.best_loss:
            self.best_loss = loss.item()
        self.log_dict({f'val/{k}': v.detach() for k, v in logs.items()}, prog_bar=True, logger=True,
                      on_step=False, on_epoch=True)
        return loss

    def configure_optimizers(self):  # :304-331 -- AdamW(1e-5) + cosine annealing to lr/100, per step
        params = []
        if self.encoder.use_dynamic_ops:
            params.extend(self.encoder.conv_in.parameters())
        if self.decoder.use_dynamic_ops:
            params.extend(self.decoder.conv_out.parameters())
        if not params:
            raise ValueError('No dynamic layers found! Make sure use_dynamic_ops=True '
                             'in encoder and/or decoder config.')
        optimizer = FusedAdam(params, lr=self.config.lr, weight_decay=1e-5)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=self.config.max_steps,
                                                               eta_min=self.config.lr * 0.01)
        return {'optimizer': optimizer, 'lr_scheduler': {'scheduler': scheduler, 'interval': 'step'}}


class DummyDataset(torch.utils.data.Dataset):  # :339-349
    def __init__(self, length: int = 10000):
        self.length = length

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return {'dummy': torch.tensor(0)}


class DummyDataModule:  # :352-363
    def __init__(self, train_size: int = 10000, val_size: int = 100):
        self.train_size, self.val_size = train_size, val_size

    def train_dataloader(self):
        return torch.utils.data.DataLoader(DummyDataset(self.train_size), batch_size=1, num_workers=0)

    def val_dataloader(self):
        return torch.utils.data.DataLoader(DummyDataset(self.val_size), batch_size=1, num_workers=0)


def create_experiment_dir(config: dict, suffix: str = 'distill') -> dict:  # :371-391
    os.makedirs(config['experiment']['exp_dir'], exist_ok=True)
    exp_name = f'{config["experiment"]["experiment_name"]}_{suffix}_{datetime.now().strftime("%m-%d-%Y_%H-%M-%S")}'
    exp_dir = os.path.join(config['experiment']['exp_dir'], exp_name)
    os.makedirs(exp_dir, exist_ok=True)
    config['experiment']['experiment_name'] = exp_name
    config['experiment']['save_dir'] = exp_dir
    print(f'Experiment directory: {exp_dir}')
    return config


def save_distillation_checkpoint(module: DistillationModule, save_path: str, config: dict) -> None:
    """The `.pt` layout EOFluxVAE(ckpt_path=...) recognises as a distilled checkpoint (:394-434)."""
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}  # noqa: E731
    checkpoint = {
        'encoder_conv_in_state_dict': cpu(module.encoder.conv_in.state_dict()) if module.encoder.use_dynamic_ops else None,
        'decoder_conv_out_state_dict': cpu(module.decoder.conv_out.state_dict()) if module.decoder.use_dynamic_ops else None,
        'encoder_state_dict': cpu(module.encoder.state_dict()),
        'decoder_state_dict': cpu(module.decoder.state_dict()),
        'distill_config': {'rgb_wavelengths': module.config.rgb_wavelengths, 'final_loss': module.best_loss,
                           'max_steps': module.config.max_steps, 'lr': module.config.lr},
        'teacher_shapes': {'encoder_weight': tuple(module.teacher_enc_w.shape),
                           'decoder_weight': tuple(module.teacher_dec_w.shape)},
        'model_config': config.get('model') if isinstance(config, dict) else None,
    }
    torch.save(checkpoint, save_path)
    print(f'Distillation checkpoint saved to: {save_path}')


def load_distilled_checkpoint(encoder: nn.Module, decoder: nn.Module, ckpt_path: str, strict: bool = True) -> dict:
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)  # :437-468
    if encoder.use_dynamic_ops and ckpt.get('encoder_conv_in_state_dict'):
        missing, unexpected = encoder.conv_in.load_state_dict(ckpt['encoder_conv_in_state_dict'], strict=strict)
        print(f'Loaded encoder.conv_in: {len(missing)} missing, {len(unexpected)} unexpected')
    if decoder.use_dynamic_ops and ckpt.get('decoder_conv_out_state_dict'):
        missing, unexpected = decoder.conv_out.load_state_dict(ckpt['decoder_conv_out_state_dict'], strict=strict)
        print(f'Loaded decoder.conv_out: {len(missing)} missing, {len(unexpected)} unexpected')
    print(f'Distillation loss was: {ckpt["distill_config"]["final_loss"]:.6f}')
    return ckpt


def run_distillation(config: dict, teacher_ckpt_path: str, distill_config: DistillConfig) -> str:  # :476-580
    from .trainer import Trainer
    print('=' * 70 + '\nEO-VAE Weight Distillation\n' + '=' * 70)
    teacher_weights = load_teacher_weights(teacher_ckpt_path)
    print('\nInstantiating encoder and decoder from config...')
    encoder, decoder = instantiate(config['model']['encoder']), instantiate(config['model']['decoder'])
    if not encoder.use_dynamic_ops and not decoder.use_dynamic_ops:
        raise ValueError('Neither encoder nor decoder uses dynamic ops. Nothing to distill!')
    print(f'  Encoder dynamic ops: {encoder.use_dynamic_ops}\n  Decoder dynamic ops: {decoder.use_dynamic_ops}')
    module = DistillationModule(encoder, decoder, teacher_weights, distill_config)
    data = DummyDataModule(train_size=distill_config.max_steps, val_size=100)
    save_dir = config['experiment']['save_dir']
    trainer = Trainer(max_steps=distill_config.max_steps, max_epochs=1, default_root_dir=save_dir,
                      val_check_interval=distill_config.val_every_n_steps, limit_val_batches=1,
                      log_every_n_steps=distill_config.log_every_n_steps, enable_checkpointing=False)
    import yaml
    with open(os.path.join(save_dir, 'config.yaml'), 'w') as f:
        yaml.safe_dump(config, f)
    print('\nStarting distillation training...')
    trainer.fit(module, train_dataloaders=data.train_dataloader(), val_dataloaders=data.val_dataloader())
    final_path = os.path.join(save_dir, 'distilled_final.pt')
    save_distillation_checkpoint(module, final_path, config)
    print('\n' + '=' * 70 + '\nDistillation Complete!\n' + '=' * 70)
    print(f'Final loss: {module.best_loss:.6f}\nCheckpoint saved to: {final_path}')
    return final_path


def main(argv=None):  # :588-645
    import argparse
    from .config import load_config
    ap = argparse.ArgumentParser(description='EO-VAE Weight Distillation',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument('--config', type=str, required=True, help='Path to model config YAML')
    ap.add_argument('--teacher-ckpt', type=str, required=True,
                    help='Path to pretrained Flux VAE checkpoint (.safetensors or .ckpt)')
    ap.add_argument('--max-steps', type=int, default=5000, help='Maximum training steps')
    ap.add_argument('--lr', type=float, default=1e-4, help='Learning rate')
    ap.add_argument('--val-every', type=int, default=500, help='Validation frequency (steps)')
    ap.add_argument('--patience', type=int, default=10, help='Early stopping patience')
    ap.add_argument('--exp-name', type=str, default=None, help='Override experiment name')
    args = ap.parse_args(argv)
    config = load_config(args.config)
    if args.exp_name:
        config['experiment']['experiment_name'] = args.exp_name
    config = create_experiment_dir(config, suffix='distill')
    dcfg = DistillConfig(max_steps=args.max_steps, lr=args.lr, val_every_n_steps=args.val_every,
                         patience=args.patience)
    return run_distillation(config, args.teacher_ckpt, dcfg)
