"""Diagnosis: do one-pass GroupNorm backward time-outs occur when the host runs ahead of the GPU (no per-step sync)?
Prints the steps whose event time is far off the median, the recorded time-outs and whether the loss stays finite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['EOVAE_GN_TIMEOUT_ACTION'] = 'warn'
import torch, bench
from eo_vae_amd import ops
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
from eo_vae_amd.dynamic_conv import prefer_rocblas_for_generator
prefer_rocblas_for_generator()
dev = torch.device('cuda', 0)
model = bench.build_model(3).to(dev); model.train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
b = {'image': synthetic_images(16, 3, 256, g).to(dev), 'wvs': torch.tensor(WAVELENGTHS['S2RGB'], device=dev), 'modality': 'S2RGB'}
for rnd in range(3):
    n = 30
    if rnd == 1:
        ops.CONV3X3_TIMER = ops.KernelTimer(); ops.WGRAD3X3_TIMER = ops.KernelTimer()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    losses = []
    ev[0].record()
    for i in range(n):
        losses.append(drv.step(b, i).detach())
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    med = sorted(ms)[n // 2]
    print(f'round {rnd} (kernel timers {"on" if rnd == 1 else "off"}): median {med:.1f} ms, max {max(ms):.1f} ms at step {ms.index(max(ms))}, '
          f'finite losses {sum(bool(torch.isfinite(l)) for l in losses)}/{n}, time-out word now {ops.L.load().eovae_gn_bwd_timeouts(0):#x}', flush=True)
    ops.CONV3X3_TIMER = None; ops.WGRAD3X3_TIMER = None
