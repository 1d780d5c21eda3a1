"""Where the GPU idles inside a train step: torch.profiler (device activities only) over 3 back-to-back cfg2 steps, then
the union of kernel intervals over all streams; prints host enqueue vs device time, the idle total, and the largest gaps
with the kernels on either side.   python tools/step_gaps.py [min_gap_us]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
from torch.profiler import profile, ProfilerActivity

min_gap = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
wl = WAVELENGTHS['S2RGB']
model = bench.build_model(3).cuda().train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
b = {'image': synthetic_images(16, 3, 256, g).cuda(), 'wvs': torch.tensor(wl).cuda(), 'modality': 'S2RGB'}
for i in range(4):
    drv.step(b, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for i in range(3):
        drv.step(b, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'host enqueue {1e3 * (t1 - t0) / 3:.1f} ms per step, wall {1e3 * (t2 - t0) / 3:.1f} ms per step')
ev = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
print('kernels', len(ev), 'span ms', (ev[-1].time_range.end - ev[0].time_range.start) / 1e3)
end, last = ev[0].time_range.end, ev[0]
gaps, idle = [], 0.0
for e in ev[1:]:
    if e.time_range.start > end:
        gap = e.time_range.start - end
        idle += gap
        if gap >= min_gap:
            gaps.append((gap, end - ev[0].time_range.start, last.name[:60], e.name[:60]))
    if e.time_range.end > end:
        end, last = e.time_range.end, e
print(f'idle total {idle / 3e3:.2f} ms per step; gaps >= {min_gap} us: {len(gaps)} ({sum(x[0] for x in gaps) / 3e3:.2f} ms per step)')
for gap, at, a, c in sorted(gaps, reverse=True)[:40]:
    print(f'  {gap:7.1f} us at {at / 1e3:8.2f} ms   {a}  ->  {c}')
# the generator kernels: where they sit and how long their neighbourhood takes
gen = [e for e in ev if 'gen_' in e.name or 'Cijk' in e.name]
print('generator GEMM-ish kernels', len(gen), 'busy us per step', sum(e.device_time for e in gen) / 3)
# small-gap histogram
import collections
h = collections.Counter()
end = ev[0].time_range.end
for e in ev[1:]:
    if e.time_range.start > end:
        gsz = e.time_range.start - end
        h[min(int(gsz // 2) * 2, 30)] += gsz
    end = max(end, e.time_range.end)
print('idle by gap size (us bucket -> ms per step):', {k: round(v / 3e3, 2) for k, v in sorted(h.items())})
# timeline dump of the middle step: boundaries = the optimizer kernel
names = [e.name for e in ev]
opt = [i for i, n in enumerate(names) if 'adam' in n.lower()]
print('optimizer kernel indices', opt)
if len(opt) >= 2:
    lo, hi = opt[0], opt[1]
    base = ev[lo].time_range.start
    out = os.environ.get('STEP_DUMP', 'gpurun_out/step_timeline.txt')
    with open(out, 'w') as f:
        for e in ev[lo:hi + 1]:
            f.write(f'{(e.time_range.start - base) / 1e3:9.3f} {e.device_time:8.1f} {e.name[:70]}\n')
    print('timeline of', hi - lo + 1, 'kernels ->', out)
