"""Kernel launches of one hyper-network forward + backward (the head and the tail of a train step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from eo_vae_amd.dynamic_conv import DynamicConv
from eo_vae_amd.data import WAVELENGTHS
m = DynamicConv(wv_planes=256, inter_dim=128, kernel_size=3, embed_dim=128, num_layers=4).cuda().train()
wvs = torch.tensor(WAVELENGTHS['S2RGB']).cuda()
for _ in range(3):
    w, b = m.get_distillation_weight(wvs)
    (w.square().sum() + b.square().sum()).backward()
torch.cuda.synchronize()
for phase in ('fwd', 'fwd+bwd'):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        w, b = m.get_distillation_weight(wvs)
        if phase != 'fwd':
            (w.square().sum() + b.square().sum()).backward()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(phase, 'kernels', len(ev), 'busy us', round(sum(e.device_time for e in ev)))
    t0 = min(e.time_range.start for e in ev)
    t1 = max(e.time_range.end for e in ev)
    print('   span us', round(t1 - t0))
    if phase == 'fwd' or os.environ.get('GEN_KERNELS_ALL'):
        for e in ev:
            print('   ', round(e.device_time, 1), e.name[:90])
# wall time of fwd + bwd with the queue kept full (what a step pays when the host runs ahead)
import time
torch.cuda.synchronize()
a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = time.perf_counter()
a.record()
for _ in range(20):
    w, b = m.get_distillation_weight(wvs)
    (w.square().sum() + b.square().sum()).backward()
b_.record()
host = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
print('fwd+bwd: device', round(a.elapsed_time(b_) / 20 * 1e3), 'us per iteration; host enqueue', round(host * 1e6), 'us')
