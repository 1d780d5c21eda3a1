"""Which host call sites launch the small kernels of a train step: torch.profiler over ONE cfg2 step (after 2 warm-ups),
kernels grouped by name with the Python stack that launched them.
    python tools/step_ops_profile.py [max_avg_us]     (default: kernels below 60 us average)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
from eo_vae_amd.dynamic_conv import prefer_rocblas_for_generator

prefer_rocblas_for_generator()
lim = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device('cuda', 0)
wl = WAVELENGTHS['S2RGB']
model = bench.build_model(len(wl)).to(dev).train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
batch = {'image': synthetic_images(16, len(wl), 256, g).to(dev), 'wvs': torch.tensor(wl, device=dev), 'modality': 'S2RGB'}
for i in range(2):
    drv.step(batch, i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    drv.step(batch, 2)
    torch.cuda.synchronize()
ev = prof.events()
# kernel events carry no stack; the launching CPU op does: walk linked kernels
by = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        stack = [s for s in (e.stack or []) if 'eo_vae_amd' in s or 'bench.py' in s]
        site = stack[0].split('/')[-1] if stack else '?'
        for k in e.kernels:
            b = by[k.name[:70]]
            b[0] += 1
            b[1] += k.duration
            b[2][(e.name, site + ' ' + str(e.input_shapes)[:90])] += 1
aten = collections.Counter(); streams = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            aten['aten' if e.name.startswith('aten::') else 'own'] += k.duration
            streams[getattr(k, 'device_resource_id', getattr(k, 'thread', '?'))] += k.duration
byop = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and e.name.startswith('aten::'):
        key = (e.name, str(e.input_shapes)[:100])
        byop[key][0] += len(e.kernels)
        byop[key][1] += sum(k.duration for k in e.kernels)
print('aten ops by kernel time:')
for (op, shp), (n, t) in sorted(byop.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'  {t / 1e3:7.3f} ms {n:4d} kernels  {op:34s} {shp}')
print('kernel time launched by aten ops vs own autograd functions (us):', dict(aten))
print('kernel time per stream (us):', dict(streams))
tot = sum(b[1] for b in by.values())
print(f'kernels linked to CPU ops: {sum(b[0] for b in by.values())} launches, {tot / 1e3:.2f} ms')
small = {k: b for k, b in by.items() if b[1] / b[0] <= lim}
print(f'below {lim:.0f} us average: {sum(b[0] for b in small.values())} launches, {sum(b[1] for b in small.values()) / 1e3:.2f} ms')
for k, b in sorted(small.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{b[1] / 1e3:7.3f} ms {b[0]:5d} x {b[1] / b[0]:6.1f} us  {k}')
    for (op, site), n in b[2].most_common(6):
        print(f'            {n:4d}  {op:28s} {site}')
sys.stdout.flush()
