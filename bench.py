#!/usr/bin/env python
"""Headline benchmark: EOFluxVAE train-step images/sec on synthetic [B,C,256,256] patches.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = EOFluxVAE.training_step on one batch already resident in HBM: dynamic-conv
hyper-network, encoder, posterior sample (noise drawn on the CPU generator like the reference),
latent norm, decoder, L1 + 1e-6*KL, backward, (bucketed RCCL all-reduce when N > 1), global
grad-norm clip at 1.0, Adam, LR schedule.  Workload = BASELINE.json configs[1]:
S2RGB 3-ch 256x256, batch 16 per GPU (weak scaling: every rank builds its own batch).
Rank 0 prints ONE JSON line."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image of the train step (forward x 3)
F_STEP_PER_IMG = {3: 2.687e12, 12: 2.695e12}
Q_STEP_PER_IMG = 9.90e9          # bytes / image
Q_STEP_FIXED = 4.8e9             # bytes / step (weights, Adam, clip)
PEAK_FP32_MFMA = 157.3e12        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA = 2500e12         # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_HBM = 8.0e12


def csrc_digest():
    """sha256 over the kernel sources: identifies the build a PMC measurement belongs to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'eo_vae_amd', 'csrc', '*'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC fold (tools/pmc_traffic.sh runs the
    SEPARATE rocprofv3 --pmc passes; this run cannot measure them itself).  The number is only quoted when
    the fold was taken on THIS build of the kernels (same digest of eo_vae_amd/csrc); otherwise traffic is
    null and the source says why -- a stale file is never passed off as a measurement."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))   # by round tag (mtimes do not travel)
    from eo_vae_amd import ops
    build = csrc_digest()
    newest = None
    for f in reversed(files):   # newest tag first; the fold that was measured on THIS build and mode wins
        try:
            with open(f) as fh:
                d = json.load(fh)
        except Exception:
            continue
        k = d.get('kernels', {}).get(kernel)
        if k is None:
            continue
        src = {'file': os.path.relpath(f, ROOT), 'csrc_digest': d.get('csrc_digest'), 'build_digest': build,
               'fetch_bytes_raw': k.get('fetch_bytes_raw_per_launch'), 'write_bytes': k.get('write_bytes_per_launch'),
               'note': 'FETCH_SIZE x2 (gfx950 correction, calibrated for 16-B/lane streams; halo loads are dword/lane, '
                       'so the true figure lies between raw and corrected)'}
        prec = d.get('conv_precision', ops.DEFAULT_PRECISION)
        if prec == ops.CONV_PRECISION and d.get('csrc_digest') == build:
            return k['traffic_bytes_per_launch_corrected'], src
        if newest is None:
            src['stale'] = (f'fold measured in conv precision {prec!r}, this run is {ops.CONV_PRECISION!r}'
                            if prec != ops.CONV_PRECISION else 'kernel sources changed since this fold was measured')
            newest = src
    if newest is not None:
        return None, newest
    return None, {'file': None, 'note': 'no PMC fold for this kernel under profiles/'}


def wgrad_roofline(wtimer, peak, elapsed, prec):
    """Second-largest kernel family of the step (the 3x3 weight gradients), measured the same way as `roofline`:
    algorithmic flops of every launch / HIP-event time around it on the launch stream."""
    ms, flops, n = wtimer.totals()
    if n == 0 or ms <= 0:
        return None
    kernel = {'f16c8': 'wgrad3x3_f16c8_rows_kernel<0|1|2> (row streaming: plain layers with record input | x2-upsample layers, '
                       'sub-pixel form | stride-2 layers, swapped operands; each incl. its records / fold launches) + '
                       'wgrad3x3_f16c8_kernel (fp32-input plain layers)',
              'f32': 'conv_wgrad_kernel'}.get(prec, 'wgrad3x3_bf16_kernel')
    ach = flops / (ms * 1e-3)
    return {'bound': 'mfma', 'kernel': kernel + ' -- algorithmic flops as the reference executes each layer (9 taps on its output grid)',
            'achieved': round(ach / 1e12, 2), 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'launches': n, 'achieved_per_launch_ms': round(ms / n, 4),
            'kernel_time_share': round(ms / (elapsed * 1e3), 4)}


def traffic_ratios(src, alg_bytes):
    """PMC traffic of the dominant kernel against its algorithmic bytes, with the raw FETCH_SIZE and with the gfx950
    x2 correction (calibrated for 16-byte-per-lane streams; dword-per-lane loads lie between the two)."""
    try:
        return {'raw': round((src['fetch_bytes_raw'] + src['write_bytes']) / alg_bytes, 3),
                'corrected': round((2 * src['fetch_bytes_raw'] + src['write_bytes']) / alg_bytes, 3)}
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def one_wave_roofline(timer, kind, peak, elapsed):
    """The other of the two one-wave-per-SIMD kernels (kind 1: conv_f16c8_w4.hip, 2: conv_f16c8_q16.hip), measured like `roofline`."""
    ms, flops, n = timer.totals(kind)
    if n == 0 or ms <= 0:
        return None
    ach = flops / (ms * 1e-3)
    name = {1: 'conv3x3_f16c8_w4_kernel (32x32 MFMAs: data gradients, fp32 / format-1 inputs)',
            2: 'conv3x3_f16c8_q16_kernel (16x16 MFMAs, all-DMA feed: forward on MX operand records)'}[kind]
    return {'bound': 'mfma', 'kernel': name, 'achieved': round(ach / 1e12, 2), 'peak': peak / 1e12, 'unit': 'TFLOP/s',
            'frac': round(ach / peak, 4), 'launches': n, 'achieved_per_launch_ms': round(ms / n, 4),
            'kernel_time_share': round(ms / (elapsed * 1e3), 4)}


def family_roofline(timer, peak, elapsed):
    ms = timer.totals(1)[0] + timer.totals(2)[0]
    flops = timer.totals(1)[1] + timer.totals(2)[1]
    n = timer.totals(1)[2] + timer.totals(2)[2]
    if n == 0 or ms <= 0:
        return None
    ach = flops / (ms * 1e-3)
    return {'kernels': 'conv3x3_f16c8_w4_kernel + conv3x3_f16c8_q16_kernel', 'achieved': round(ach / 1e12, 2),
            'frac': round(ach / peak, 4), 'launches': n, 'kernel_time_share': round(ms / (elapsed * 1e3), 4)}


def up_roofline(timer, peak, elapsed):
    """The x2-upsample forward / data-gradient launches on conv_f16c8_q16up.hip / conv_f16c8_q16upg.hip (kind 3), measured like
    `roofline`; algorithmic flops = the layer as the reference executes it (9 taps on the 2x grid; the sub-pixel forms do 4/9 of
    that in MFMAs)."""
    ms, flops, n = timer.totals(3)
    if n == 0 or ms <= 0:
        return None
    ach = flops / (ms * 1e-3)
    return {'bound': 'mfma', 'kernel': 'conv3x3_f16c8_q16up_kernel (x2-upsample forward on MX records of the low-res input, both column '
                                       'parities per wave) + conv3x3_f16c8_q16upg_kernel (its data gradient on MX records of dY, K over the '
                                       'four parity planes), each incl. its records_mx_kernel pass',
            'achieved': round(ach / 1e12, 2), 'achieved_actual_mfma_work': round(ach * 4.0 / 9.0 / 1e12, 2), 'peak': peak / 1e12,
            'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'launches': n, 'achieved_per_launch_ms': round(ms / n, 4),
            'kernel_time_share': round(ms / (elapsed * 1e3), 4)}


def conv8_roofline(timer, peak, elapsed):
    """The 3x3 forward / data-gradient launches that stay on the 8-wave kernel (sub-pixel and space-to-depth forms, grids
    below two 16-row tiles per CU), measured like `roofline`."""
    ms, flops, n = timer.totals(0)
    if n == 0 or ms <= 0:
        return None
    ach = flops / (ms * 1e-3)
    return {'bound': 'mfma', 'kernel': 'conv3x3_f16c8_kernel (8 waves: sub-pixel / space-to-depth forms and small grids)',
            'achieved': round(ach / 1e12, 2), 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'launches': n, 'achieved_per_launch_ms': round(ms / n, 4), 'kernel_time_share': round(ms / (elapsed * 1e3), 4)}


def build_model(channels, kl_weight=1e-6, msssim=False):
    from eo_vae_amd.autoencoder import EOFluxVAE
    from eo_vae_amd.losses import EOConsistencyLoss
    from eo_vae_amd.model import Decoder, Encoder
    dyn = {'num_layers': 4, 'wv_planes': 256}
    torch.manual_seed(0)
    enc = Encoder(resolution=256, in_channels=channels, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                  z_channels=32, use_dynamic_ops=True, dynamic_conv_kwargs=dict(dyn))
    dec = Decoder(ch=128, out_ch=channels, ch_mult=[1, 2, 4, 4], num_res_blocks=2, resolution=256,
                  z_channels=32, use_dynamic_ops=True, dynamic_conv_kwargs=dict(dyn))
    loss = EOConsistencyLoss(pixel_weight=1.0, rec_loss_type='l1', kl_weight=kl_weight,
                             msssim_weight=1.0 if msssim else 0.0)
    # configs/eo-vae.yaml:15-23 optimisation settings
    return EOFluxVAE(enc, dec, loss, freeze_body=False, base_lr=1e-4, final_lr=2e-5, warmup_epochs=1,
                     decay_end_epoch=100, clip_grad=1.0)


class StepDriver:
    """The slice of eo_vae_amd.trainer.Trainer that one training step needs."""

    def __init__(self, model, world):
        from eo_vae_amd.parallel import BucketedGradReducer
        conf = model.configure_optimizers()
        opts, scheds = conf if isinstance(conf, tuple) else (conf, [])
        self.opt = opts[0]
        model.attach(self, opts, [s['scheduler'] for s in scheds])
        self.reducer = None
        self.buffers = [b for b in model.buffers() if b.is_floating_point()]
        if world > 1:
            self.reducer = BucketedGradReducer(self.opt.flat)
            self.reducer.broadcast_parameters(self.buffers)
        self.model = model

    def log_metrics(self, metrics, **kw):
        pass

    def step(self, batch, i):
        if self.reducer is not None:
            for b in self.buffers:
                dist.broadcast(b, src=0)
        loss = self.model.training_step(batch, i)
        self.model._global_step += 1
        return loss


def host_cpu():
    """(model name, physical cores, logical CPUs this process may use) of the box's host."""
    model, phys = 'unknown', set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                pid = v
            elif k == 'core id':
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return model, (len(phys) or logical), logical


def cpu_baseline(channels, wvs_list, micro_batch=2):
    """Oracle (CPU restatement, kind 'port') timed on this box's host cores: full train steps at a
    bounded micro-batch (B=16 needs ~56 GB of saved activations on the host) with torch CPU threads = the physical
    cores this process may use, 32 (where the oracle's convolutions stop scaling) and 8 (the build container's
    count).  `value` / `cores` = the fastest of the three; all three are in `by_threads`."""
    from oracle import eo_vae_cpu as O
    model_name, phys, logical = host_cpu()
    m = O.build_vae(O.model_cfg(channels=channels))
    O.det_fill(m, seed=31)
    m.train()
    st = O.TrainState(m, base_lr=1e-4, final_lr=2e-5, warmup_epochs=1, decay_end_epoch=100, clip_grad=1.0)
    loss_fn = O.EOConsistencyLoss(pixel_weight=1.0, rec_loss_type='l1', kl_weight=1e-6)
    wvs = torch.tensor(wvs_list)
    x = O.synthetic_batch(micro_batch, channels, 256, seed=99)
    # 1 warm-up + 3 timed steps at the thread count that is reported as `value`.  Which count that is has to be
    # found on the box: one warm-up + one timed step each at the physical cores this process may use, at 32 (where the
    # oracle's convolutions stop scaling) and at 8 (the build container's count), then three timed steps at the fastest.
    cands = []
    for thr in (max(1, min(phys, logical)), 32, 8):
        thr = max(1, min(thr, logical))
        if thr not in cands:
            cands.append(thr)
    by = {}
    for thr in cands:
        torch.set_num_threads(thr)
        st.step(loss_fn, x, wvs)            # warm-up at this thread count (thread pool, oneDNN primitive caches)
        t0 = time.time()
        st.step(loss_fn, x, wvs)
        dt = time.time() - t0
        by[str(thr)] = {'images_per_sec': round(micro_batch / dt, 4), 's_per_step': round(dt, 2), 'timed_steps': 1}
    best = max(by, key=lambda k: by[k]['images_per_sec'])
    torch.set_num_threads(int(best))
    st.step(loss_fn, x, wvs)                # warm-up again: the pool was resized in between
    times = []
    for _ in range(3):
        t0 = time.time()
        st.step(loss_fn, x, wvs)
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    by[best] = {'images_per_sec': round(micro_batch / dt, 4), 's_per_step': round(dt, 2), 'timed_steps': 3,
                's_per_step_each': [round(t, 2) for t in times]}
    return {'value': by[best]['images_per_sec'], 'unit': 'images/sec', 'cores': int(best), 'kind': 'port',
            'cpu_model': model_name, 'physical_cores': phys, 'logical_cpus': logical, 'by_threads': by,
            'sample': f'1 warm-up + 3 timed full train steps (fwd+L1+KL+bwd+clip+Adam) at micro-batch {micro_batch}, '
                      f'{channels}x256x256, fp32, torch CPU, {best} threads ({by[best]["s_per_step"]} s/step); the thread count '
                      f'was chosen from one timed step each at {", ".join(str(c) for c in cands)} threads (by_threads)'}


# Shader clock the chip holds under a dense matrix load: MEASURED in this run by eovae_clock_probe right behind the timed
# region (s_memtime / s_memrealtime around an MFMA loop on every CU, median over workgroups) -- what `ceiling` prices
# the matrix cores at.  (Round 3 used a constant, 1.77 GHz from a diagnostic build; boxes of the pool run the same build
# at 91 or 101 ms per step depending on a clock state that the constant hid.)
MAX_CLOCK_GHZ = 2.4
CLOCK_STATE_HIGH_GHZ = 1.95   # probe readings at or above this: the box holds its higher clock state


def clock_probe(device, ms=40.0):
    """Median over workgroups of d(s_memtime) / d(s_memrealtime) under eovae_clock_probe's MFMA loop, in GHz; the loop
    runs ~`ms` milliseconds after one untimed launch of the same length (the chip settles into its held clock)."""
    import numpy as np
    from eo_vae_amd import lib as L
    lib = L.load()
    blocks = torch.cuda.get_device_properties(device).multi_processor_count
    out = torch.zeros(2 * blocks, dtype=torch.int64, device=device)
    sink = torch.zeros(1, device=device)
    iters = max(1, int(ms * 1e-3 / 0.55e-6))     # 16 MFMAs x 32 cycles x 2 waves per SIMD ~ 0.55 us at 1.9 GHz
    for _ in range(2):
        L.check(lib.eovae_clock_probe(L.ptr(out), L.ptr(sink), blocks, iters, L.stream()), 'eovae_clock_probe')
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype(np.float64).reshape(blocks, 2)
    ghz = np.median(v[:, 0] / np.maximum(v[:, 1], 1.0)) * 0.1
    return float(ghz), float(np.median(v[:, 1]) * 1e-5)     # (GHz, loop length in ms)
# GroupNorm moves 1.405 GB/img forward, x3 for the step; optimiser + clip + weights 4.8 GB/step
Q_GN_STEP_PER_IMG = 3 * 1.405e9


def f16c8_units(timer, mx_on, dgrad_mx):
    """MFMA units per algorithmic product of the timed 3x3 forward / data-gradient launches, weighted by their flops."""
    u = {0: 2.0, 1: 1.64 if (mx_on and dgrad_mx) else 2.0, 2: 1.64 if mx_on else 2.0, 3: 1.64}
    fl = {k: timer.totals(k)[1] for k in u}
    tot = sum(fl.values())
    return sum(fl[k] * u[k] for k in u) / tot if tot > 0 else 2.0


def ceiling(conv_flops_step, wgrad_flops_step, batch, ips, units=2.0, clock_ghz=1.77, units_wgrad=None):
    """How far the north_star's 40 % of the HBM-roofline rate is, from this line alone.
    matrix-core floor = algorithmic 3x3 flops of the step (forward + data gradient + weight gradient, as timed above)
    x MFMA units per product (f16c8: 1.64 on the MX-FP6 kernels -- forward on operand records, data gradients, the x2-upsample
    kernels --, 2 on the 8-wave kernel's fp32-input forms, 1 in the f16 weight gradients; flops-weighted, f16c8_units) / dense peak
    at the clock the chip HOLDS under this load;
    HBM floor = the passes no matrix kernel can hide at the HBM peak: GroupNorm (forward apply, backward reduce +
    apply: SURVEY's 3 x 1.405 GB/img) and optimiser / clip / weight traffic (4.8 GB/step);
    1x1 / attention GEMMs, losses, the hyper-network and launch gaps are NOT in the floor (they are ~10 ms today)."""
    units_wgrad = units if units_wgrad is None else units_wgrad
    work = conv_flops_step * units + wgrad_flops_step * units_wgrad      # flops x units
    peak_hold = PEAK_BF16_MFMA * clock_ghz / MAX_CLOCK_GHZ
    t_mfma_hold = work / peak_hold
    t_mfma_max = work / PEAK_BF16_MFMA
    t_hbm = (Q_GN_STEP_PER_IMG * batch + Q_STEP_FIXED) / PEAK_HBM
    r_hbm = PEAK_HBM / (Q_STEP_PER_IMG + Q_STEP_FIXED / batch)
    out = {
        'mfma_units_per_product': {'forward_and_data_gradient': round(units, 3), 'weight_gradient': round(units_wgrad, 3)},
        'clock_under_load_ghz': round(clock_ghz, 3),
        'clock_source': 'eovae_clock_probe in this run (s_memtime / s_memrealtime around an f16-MFMA loop on every CU, right behind the timed region)',
        'matrix_floor_ms_at_held_clock': round(t_mfma_hold * 1e3, 2), 'matrix_floor_ms_at_2.4ghz': round(t_mfma_max * 1e3, 2),
        'hbm_floor_ms_unfused_passes': round(t_hbm * 1e3, 2),
        'ceiling_ms_per_step': round((t_mfma_hold + t_hbm) * 1e3, 2),
        'ceiling_images_per_sec': round(batch / (t_mfma_hold + t_hbm), 1),
        'ceiling_hbm_roofline_frac': round(batch / (t_mfma_hold + t_hbm) / r_hbm, 4),
        'target_hbm_roofline_frac': 0.40, 'target_images_per_sec': round(0.40 * r_hbm, 1),
        'achieved_frac_of_ceiling': round(ips / (batch / (t_mfma_hold + t_hbm)), 4),
        'note': 'at 100 % MFMA-busy and the clock the chip holds, plus the GroupNorm / optimiser passes at the HBM peak; the '
                'target needs ~51 ms/step: ' + ('inside this ceiling' if (t_mfma_hold + t_hbm) * 1e3 < 51.0 else
                                                'beyond this ceiling, i.e. out of reach for this arithmetic at this clock')
                + ' (1x1 / attention GEMMs, losses, the hyper-networks and launch gaps are not in it)',
    }
    return out


def parity_vs_reference():
    """BASELINE.json's second half of the metric: error of the GPU path against the REFERENCE's CPU output
    on config 1 (full-size model, S2RGB 3x256x256, batch 2, reconstruct()), from the committed fixture
    tests/golden/full_cfg1.npz that tests/golden/gen_golden.py produced by running the reference itself.
    Part of the checker leg (imports oracle/ for the closed-form parameter fill and the seeded input)."""
    import numpy as np
    from oracle import eo_vae_cpu as O
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'full_cfg1.npz'))
    model = build_model(3)
    O.det_fill(model, seed=31)          # the rule the fixture's reference model was filled with
    model = model.cuda().eval()
    wvs = torch.tensor(O.WAVELENGTHS['S2RGB']).cuda()
    x = O.synthetic_batch(2, 3, 256, seed=1234)
    assert abs(float(x.double().sum()) - float(g['x_checksum'])) < 1e-6 * abs(float(g['x_checksum'])) + 1e-6
    with torch.no_grad():
        rec = model.reconstruct(x.cuda(), wvs)[:, :, ::8, ::8].double().cpu()
        lat = model.encode_spatial_normalized(x.cuda(), wvs)[:, :, ::2, ::2].double().cpu()
    ref_r, ref_l = torch.from_numpy(g['recon_sub']).double(), torch.from_numpy(g['spatial_sub']).double()
    mse = float(((rec - ref_r) ** 2).mean())
    rng = float(ref_r.max() - ref_r.min())
    # the train step itself at full size against the reference's forward + backward (fixture full_cfg2_step)
    from oracle.parity_checks import full_step_errors
    step = full_step_errors(model)
    return {'workload': 'cfg1: reconstruct(), S2RGB 3x256x256, batch 2, full-size model, deterministic fill',
            'train_step': {'workload': 'full-size model, B=2 S2RGB 256x256, train-mode forward (injected noise) + '
                                       'Charbonnier + 1e-6 KL + backward vs the REFERENCE (tests/golden/full_cfg2_step.npz)',
                           **{k: (round(v, 9) if isinstance(v, float) else v) for k, v in step.items()}},
            'train_loss_rel_err': step['loss_rel_err'], 'grad_worst_rel_err': step['grad_worst_norm_rel_err'],
            'reference': 'reference CPU fp32 output (tests/golden/full_cfg1.npz, every 8th pixel / 2nd latent)',
            'recon_max_rel_err': float((rec - ref_r).abs().max() / ref_r.abs().max()),
            'latent_max_rel_err': float((lat - ref_l).abs().max() / ref_l.abs().max()),
            'recon_psnr_db': float(10.0 * math.log10(rng * rng / mse)) if mse > 0 else float('inf'),
            'tolerance': 1e-3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch (weak scaling)')
    ap.add_argument('--modality', default=None)
    ap.add_argument('--workload', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4', 'cfg5'],
                    help='cfg2 (default, headline): S2RGB train step L1+KL; cfg3: S2L2A 12-ch train step '
                         'L1+MS-SSIM+KL; cfg4: mixed S2L2A / S1RTC random collate, batch 32 per rank (every rank '
                         'draws its own modality per step); cfg5: S2L2A encode-only (encode_spatial_normalized + '
                         'running stats)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from eo_vae_amd import ops
    from eo_vae_amd.data import WAVELENGTHS, synthetic_images
    from eo_vae_amd.dynamic_conv import prefer_rocblas_for_generator
    from eo_vae_amd.parallel import init_distributed
    blas_backend = prefer_rocblas_for_generator()   # this process is the benchmark's own
    timing_only = None
    if os.environ.get('EOVAE_TIMING_FREEZE_GENERATORS') == '1':    # what the hyper-networks cost a step: an INVALID line, marked so
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
        import generator_freeze
        generator_freeze.install()
        timing_only = 'weight generators frozen (no generator gradients): NOT a valid benchmark line'

    rank, world, local = init_distributed()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    if args.modality is None:
        args.modality = 'S2RGB' if args.workload == 'cfg2' else 'S2L2A'
    if args.workload == 'cfg5' and args.batch == 16:
        args.batch = 64
    if args.workload == 'cfg4' and args.batch == 16:
        args.batch = 32          # Lightning semantics: batch_size is per process (terramesh_datamodule.py:750)
    wl = WAVELENGTHS[args.modality]
    C = len(wl)
    model = build_model(C, msssim=args.workload == 'cfg3').to(device)
    if args.workload == 'cfg5':
        from eo_vae_amd.encode_latents import RunningStats, encode_spatial_norm
        model.eval()
        stats = RunningStats(32, device)

        class EncodeDriver:
            def step(self, batch, i):
                z = encode_spatial_norm(model, batch['image'], batch['wvs'])
                stats.update(z)
                return z.sum()
        driver = EncodeDriver()
    else:
        model.train()
        driver = StepDriver(model, world)
    g = torch.Generator('cpu').manual_seed(1234 + rank)
    nb = 2  # distinct synthetic batches resident in HBM before the timed region
    if args.workload == 'cfg4':
        # BASELINE config 4: the random collate draws ONE modality per batch per process
        # (terramesh_datamodule.py:418-503), so the ranks of one step run different channel counts.
        # One resident batch per modality; step i of rank r uses the draw of random.Random(1234 + r + i).
        import random
        mods = ['S2L2A', 'S1RTC']
        pool = {m: {'image': synthetic_images(args.batch, len(WAVELENGTHS[m]), 256, g).to(device),
                    'wvs': torch.tensor(WAVELENGTHS[m], device=device), 'modality': m} for m in mods}
        draws = [random.Random(1234 + rank + i).choice(mods) for i in range(args.warmup + args.steps)]

        class _Draw:
            def __getitem__(self, i):
                return pool[draws[self.base + i]]
        batches, nb = _Draw(), args.warmup + args.steps
        batches.base = 0
    else:
        batches = [{'image': synthetic_images(args.batch, C, 256, g).to(device),
                    'wvs': torch.tensor(wl, device=device), 'modality': args.modality} for _ in range(nb)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        driver.step(batches[i % nb], i)
    sync()
    if args.workload == 'cfg4':
        batches.base = args.warmup
    ops.CONV3X3_TIMER = ops.KernelTimer()
    ops.WGRAD3X3_TIMER = ops.KernelTimer()
    # per-step HIP events on the compute stream (SURVEY 8d timing protocol: median and p10/p90
    # beside the barrier-bracketed wall time that `value` is computed from)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = driver.step(batches[i % nb], i)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pick = lambda q: step_ms[min(len(step_ms) - 1, int(round(q * (len(step_ms) - 1))))]  # noqa: E731
    timer, ops.CONV3X3_TIMER = ops.CONV3X3_TIMER, None
    wtimer, ops.WGRAD3X3_TIMER = ops.WGRAD3X3_TIMER, None
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    final_loss = float(loss.detach())
    clock_ghz, clock_ms = clock_probe(device)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        ips = args.batch * world * args.steps / elapsed
        prec = ops.CONV_PRECISION
        # f16c8: the dominant kernel is the one-wave-per-SIMD kernel (kind 1); the launches the 8-wave kernel takes
        # (sub-pixel / space-to-depth forms, small grids) are reported beside it as `roofline_conv8`
        # (kind 2, round 5: the layers on MX operand records run conv_f16c8_q16.hip -- 16 x 16 MFMAs, all-DMA feed; the
        # dominant kernel is whichever of the two one-wave kernels holds more of the step, the other is reported beside it)
        dom = None
        if prec == 'f16c8' and (timer.totals(1)[2] > 0 or timer.totals(2)[2] > 0):
            dom = 1 if timer.totals(1)[0] >= timer.totals(2)[0] else 2
        k_ms, k_flops, k_n = timer.totals(dom)
        dom_name = {1: 'conv3x3_f16c8_w4_kernel', 2: 'conv3x3_f16c8_q16_kernel'}.get(dom, 'conv3x3_f16c8_kernel')
        traffic, traffic_source = pmc_traffic(dom_name if prec == 'f16c8' else 'conv3x3_bf16_kernel')
        achieved = k_flops / (k_ms * 1e-3) if k_ms > 0 else 0.0
        r_hbm = PEAK_HBM / (Q_STEP_PER_IMG + Q_STEP_FIXED / args.batch)  # img/s per GPU at the HBM roof
        f_img = F_STEP_PER_IMG.get(C, 2.69e12)
        if prec == 'f32':
            peak, kname, dtype = PEAK_FP32_MFMA, 'conv_fwd_kernel<3,1,8> (3x3 conv fwd + data-grad)', 'f32'
            path = 'fp32-input MFMA implicit GEMM (exact fp32)'
        elif prec == 'f16c8':
            peak, dtype = PEAK_BF16_MFMA, 'f16c8'
            kname = {1: 'conv3x3_f16c8_w4_kernel (plain 3x3 layers, one wave per SIMD, 4x4 blocks of 32x32 MFMAs: the data gradients '
                        '(fp32 inputs quantised in the kernel to f16 + MX-FP6; fp8 with c8_dgrad_mx = 0) and the forward of layers '
                        'without MX records)',
                     2: 'conv3x3_f16c8_q16_kernel (plain 3x3 layers on MX operand records -- the forward: one wave per SIMD, 8x8 '
                        'blocks of 16x16 MFMAs, MX-FP6 correction product over tap pairs, halo and weights by LDS-DMA)'
                     }.get(dom, 'conv3x3_f16c8_kernel (3x3 conv fwd + data-grad, f16 product + one fp8 correction product)')
            lib_ = ops.L.load()
            mx_on, wcorr = lib_.eovae_get_option(b'c8_mx') != 0, lib_.eovae_get_option(b'c8_wgrad_corr') != 0
            dgrad_mx = mx_on and lib_.eovae_get_option(b'c8_dgrad_mx') != 0
            path = ('f16c8 implicit GEMM: x*w = f16(x)*f16(w) + ONE K-concatenated block-scaled correction product '
                    '[x_lo|x_hi].[w_hi|w_lo], fp32 accumulate (~1e-5 per product), on v_mfma_f32_32x32x16_f16 / '
                    'v_mfma_scale_f32_32x32x64_f8f6f4 or -- forward on MX records, plain weight gradients -- on the 16x16 forms '
                    '(v_mfma_f32_16x16x32_f16 / v_mfma_scale_f32_16x16x128_f8f6f4 over tap pairs: the chip holds a higher clock on them): '
                    + ('MX-FP6 (e2m3, per-32-channel E8M0 scales; 1.64 MFMA units per product) in the forward of the layers '
                       'GroupNorm feeds with MX operand records, ' if mx_on else '')
                    + ('and -- quantised inside the one-wave kernel, a thread = a pixel x 32 channels -- in their data gradients; fp8 (e5m2 / '
                       'e4m3; 2 units) in the 8-wave kernel\'s fp32-input forms; weight gradients: ' if dgrad_mx else
                       'fp8 (e5m2 / e4m3; 2 units) in the data gradient and on fp32 inputs; weight gradients: ')
                    + ('f16 product + fp8 correction (2 units)' if wcorr else
                       'f16(dY)*f16(X) alone (1 unit: a leaf of the computation, its rounding does not compound)')
                    + '; 1x1 convolutions and attention stay split-bf16 (3 units)')
        else:
            peak = PEAK_BF16_MFMA
            kname = 'conv3x3_bf16_kernel (3x3 conv fwd + data-grad, split-bf16 operands)'
            dtype = 'bf16x3' if prec == 'bf16x3' else 'bf16'
            path = ('split-bf16 implicit GEMM: x*w = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 '
                    'accumulate (~2^-17 per product; 3 MFMAs per algorithmic one => frac <= 1/3)'
                    if prec == 'bf16x3' else 'plain bf16 operands, fp32 accumulate (fast mode, below the parity bar)')
        out = {
            'metric': 'train-step images/sec (BxCx256x256)', 'value': round(ips, 3), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype,
            'data': 'synthetic (seeded low-res Gaussian field x8 bilinear + 0.25 noise, clamp[-2.5,8]); '
                    'random-init weights',
            'config': {'workload': {'cfg2': f'cfg2: {args.modality} {C}-ch 256x256, batch {args.batch}/GPU, full train step '
                                            '(L1 + 1e-6 KL, clip 1.0, Adam, cosine-warmup LR)',
                                    'cfg3': f'cfg3: {args.modality} {C}-ch 256x256, batch {args.batch}/GPU, full train step '
                                            '(L1 + MS-SSIM + 1e-6 KL, clip 1.0, Adam)',
                                    'cfg4': f'cfg4: mixed S2L2A (12-ch) / S1RTC (2-ch) random collate, one modality per rank '
                                            f'and step, 256x256, batch {args.batch}/GPU, full train step (L1 + 1e-6 KL, '
                                            'clip 1.0, Adam)',
                                    'cfg5': f'cfg5: {args.modality} {C}-ch 256x256, batch {args.batch}/GPU, encode-only '
                                            '(encode_spatial_normalized + running latent statistics)'}[args.workload],
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}',
                       'conv_path': path, 'generator_blas': blas_backend},
            'roofline': {'bound': 'mfma', 'kernel': kname,
                         'achieved': round(achieved / 1e12, 2), 'peak': peak / 1e12,
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                         'traffic': traffic if prec != 'f32' else None, 'traffic_source': traffic_source,
                         'traffic_ratio_vs_algorithmic': traffic_ratios(traffic_source, timer.bytes_total(dom) / max(k_n, 1)),
                         'algorithmic_bytes_per_launch': round(timer.bytes_total(dom) / max(k_n, 1)),
                         'achieved_per_launch_ms': round(k_ms / max(k_n, 1), 4), 'launches': k_n,
                         'kernel_time_share': round(k_ms / (elapsed * 1e3), 4),
                         # the two one-wave-per-SIMD kernels together: the family rounds 3-4 reported as ONE kernel (their
                         # `roofline` object), for the round-over-round comparison
                         'one_wave_family': family_roofline(timer, peak, elapsed) if dom else None},
            'roofline_conv8': conv8_roofline(timer, peak, elapsed) if dom else None,
            'roofline_up': up_roofline(timer, peak, elapsed) if dom else None,
            'roofline_other_one_wave': one_wave_roofline(timer, 3 - dom, peak, elapsed) if dom else None,
            'roofline_wgrad': wgrad_roofline(wtimer, peak, elapsed, prec),
            # f16c8: MFMA units per product as a flops-weighted mean over the launches of the timed region, by the kernel that
            # took them (no hard-coded MX share): the 16 x 16 kernels on MX records and the x2-upsample kernels 1.64,
            # the one-wave 32 x 32 kernel 1.64 with the MX data gradient (2.0 without), the 8-wave kernel 2.0 (fp8 correction;
            # slightly pessimistic for its few MX-record launches)
            'ceiling': ceiling(timer.totals()[1] / args.steps, wtimer.totals()[1] / args.steps, args.batch, ips / world,
                               (f16c8_units(timer, mx_on, dgrad_mx) if prec == 'f16c8' else
                                (3.0 if prec == 'bf16x3' else (16.0 if prec == 'f32' else 1.0))), clock_ghz,
                               (2.0 if wcorr else 1.0) if prec == 'f16c8' else None)
            if args.workload != 'cfg5' else None,
            'clock_state': {'probe_ghz': round(clock_ghz, 3), 'probe_loop_ms': round(clock_ms, 1),
                            'state': 'high' if clock_ghz >= CLOCK_STATE_HIGH_GHZ else 'common',
                            'note': 'shader clock under eovae_clock_probe (dense f16 MFMAs on every CU) right behind the timed '
                                    f'region; >= {CLOCK_STATE_HIGH_GHZ} GHz = the higher of the two states boxes of this pool are '
                                    'found in (every kernel ~10 % faster there)'},
            'step_ms_events': {'median': round(pick(0.5), 3), 'p10': round(pick(0.1), 3), 'p90': round(pick(0.9), 3)},
            'step_flops_tflops': round(f_img * ips / world / 1e12, 2),
            'hbm_roofline_frac': round(ips / world / r_hbm, 4),
            'final_loss': final_loss,
        }
        if timing_only:
            out['timing_only'] = timing_only
        if args.workload == 'cfg5':
            out['metric'] = 'encode-only images/sec (BxCx256x256)'
            out.pop('hbm_roofline_frac', None)
            out['step_flops_tflops'] = round(274.6e9 * ips / world / 1e12, 2)
        if not args.no_cpu_baseline and world == 1 and args.workload == 'cfg2':
            del driver, model
            torch.cuda.empty_cache()
            out['parity'] = parity_vs_reference()
            out['cpu_baseline'] = cpu_baseline(C, wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
