#!/usr/bin/env python
"""Sharded latent encoding entry (counterpart of the reference's encode_latents.py hot loop,
:305-352): every rank encodes its own shard with the frozen VAE, running statistics are merged
across ranks with one all_gather at the end.

    python encode_latents.py --config configs/eo-vae-mi355x.yaml [--ckpt x.ckpt] --steps 8 --batch 64
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 encode_latents.py ..."""
import argparse
import os

import torch

from eo_vae_amd.autoencoder import EOFluxVAE
from eo_vae_amd.data import SyntheticLoader
from eo_vae_amd.encode_latents import encode_raw, encode_spatial_norm, encode_split, write_stats_json
from eo_vae_amd.parallel import init_distributed

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--ckpt', default=None)
    ap.add_argument('--out', default=None, help='directory for .npz latents + latent_stats.json')
    ap.add_argument('--mode', choices=['spatial_norm', 'raw'], default='spatial_norm')
    ap.add_argument('--modality', default='S2L2A')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--size', type=int, default=256)
    args = ap.parse_args()
    rank, world, local = init_distributed()
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    model = EOFluxVAE.from_config(args.config, args.ckpt, device=device, eval_mode=True)
    if args.out and rank == 0:
        os.makedirs(args.out, exist_ok=True)
    loader = SyntheticLoader([args.modality], args.batch, args.size, args.steps, 1234 + rank, device,
                             mode=args.modality)
    fn = encode_spatial_norm if args.mode == 'spatial_norm' else encode_raw
    stats = encode_split(model, loader, fn, args.out, rank)
    stats.merge_across_ranks()
    if rank == 0:
        d = stats.get_stats_dict()
        print('count', int(d['count'][0]), 'mean[:4]', d['mean'][:4].tolist(), 'std[:4]', d['std'][:4].tolist())
        if args.out:
            write_stats_json(stats, os.path.join(args.out, 'latent_stats.json'))
