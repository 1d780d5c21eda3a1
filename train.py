#!/usr/bin/env python
"""Training entry of the MI355X hot path -- counterpart of the reference's train.py (:65-208).

    python train.py --config configs/eo-vae-mi355x.yaml [--distilled-ckpt x.pt] [--vae-ckpt y.ckpt] [--debug]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config ...

Accepts the reference's YAML layout (`model:`/`datamodule:`/`trainer:` with `_target_`s).  Without
`lightning` installed the `lightning.Trainer` target resolves to eo_vae_amd.trainer.Trainer and the
TerraMesh datamodule target to the synthetic datamodule (eo_vae_amd/config.py TARGET_ALIASES)."""
from __future__ import annotations

import argparse
import os
from datetime import datetime

import torch

from eo_vae_amd.config import instantiate, load_config


def create_experiment_dir(config: dict) -> dict:
    base = config['experiment']['exp_dir']
    os.makedirs(base, exist_ok=True)
    name = f'{config["experiment"]["experiment_name"]}_{datetime.now().strftime("%m-%d-%Y_%H-%M-%S-%f")}'
    path = os.path.join(base, name)
    os.makedirs(path, exist_ok=True)
    config['experiment'].update(experiment_name=name, save_dir=path)
    config['trainer']['default_root_dir'] = path
    return config


def load_component(component, prefix: str, state_dict: dict) -> None:
    sub = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    if not sub:
        print(f'Warning: No weights found for {prefix[:-1]}')
        return
    keys = component.load_state_dict(sub, strict=False)
    print(f'Loaded {prefix[:-1]}: Missing {len(keys.missing_keys)}, Unexpected {len(keys.unexpected_keys)}')


def run_experiment(config: dict, distilled_ckpt=None, vae_ckpt=None, debug=False, overrides=None):
    print('Instantiating Encoder & Decoder...')
    encoder = instantiate(config['model']['encoder'])
    decoder = instantiate(config['model']['decoder'])
    if distilled_ckpt is not None and vae_ckpt is None:
        print(f'--- Loading Distilled Weights from {distilled_ckpt} ---')
        ckpt = torch.load(distilled_ckpt, map_location='cpu')
        sd = ckpt.get('state_dict', ckpt)
        load_component(encoder, 'encoder.', sd)
        load_component(decoder, 'decoder.', sd)
    print('Instantiating Loss Function...')
    loss_fn = instantiate(config['model']['loss_fn'])
    print('Instantiating EOFluxVAE...')
    model = instantiate(config['model'], encoder=encoder, decoder=decoder, loss_fn=loss_fn)
    if vae_ckpt:
        model._load_checkpoint(vae_ckpt, [])
    datamodule = instantiate(config['datamodule'])
    trainer_cfg = dict(config['trainer'])
    trainer_cfg.update(overrides or {})
    if debug:
        trainer_cfg.pop('default_root_dir', None)
    trainer = instantiate(trainer_cfg)
    trainer.fit(model, datamodule=datamodule)
    return trainer, model


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=str, required=True, help='Path to the config file')
    ap.add_argument('--distilled-ckpt', type=str, default=None, help='Path to distilled checkpoint')
    ap.add_argument('--vae-ckpt', type=str, default=None, help='Path to full VAE checkpoint')
    ap.add_argument('--debug', action='store_true', help='no logging, no experiment directory')
    ap.add_argument('--max-steps', type=int, default=-1)
    args = ap.parse_args()
    cfg = load_config(args.config)
    if not args.debug and int(os.environ.get('RANK', '0')) == 0:
        cfg = create_experiment_dir(cfg)
    run_experiment(cfg, args.distilled_ckpt, args.vae_ckpt, args.debug,
                   overrides={'max_steps': args.max_steps} if args.max_steps > 0 else None)
