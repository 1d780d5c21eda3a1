#!/usr/bin/env python
"""Stage-1 weight distillation entry point (same CLI as the reference's weight_distill_train.py):

    python weight_distill_train.py --config configs/eo-vae-mi355x.yaml --teacher-ckpt ae.safetensors \
        [--max-steps 5000 --lr 1e-4 --val-every 500 --exp-name NAME]
"""
from eo_vae_amd.distill import (DistillConfig, DistillationModule, load_distilled_checkpoint,  # noqa: F401
                                load_teacher_weights, main, run_distillation, save_distillation_checkpoint)

if __name__ == '__main__':
    main()
