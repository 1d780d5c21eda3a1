"""Every 3x3 weight-gradient launch of one cfg2 train step with its shape and its time (HIP events), slowest first.
    python tools/wgrad_calls.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eo_vae_amd import ops
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
from eo_vae_amd.dynamic_conv import prefer_rocblas_for_generator

prefer_rocblas_for_generator()
dev = torch.device('cuda', 0)
wl = WAVELENGTHS['S2RGB']
model = bench.build_model(len(wl)).to(dev).train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
batch = {'image': synthetic_images(16, len(wl), 256, g).to(dev), 'wvs': torch.tensor(wl, device=dev), 'modality': 'S2RGB'}
for i in range(2):
    drv.step(batch, i)
torch.cuda.synchronize()
calls = []
orig = ops._launch_wgrad3x3


def timed(x, dy, dwp, db, n, cin, h, w, cout, ho, wo, mode, subpix, packed=False, xamax=None, dw=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(x, dy, dwp, db, n, cin, h, w, cout, ho, wo, mode, subpix, packed=packed, xamax=xamax, dw=dw)
    e1.record()
    calls.append((e0, e1, cin, cout, h, w, ho, wo, subpix, packed))


ops._launch_wgrad3x3 = timed
drv.step(batch, 2)
torch.cuda.synchronize()
rows = sorted(((e0.elapsed_time(e1), cin, cout, h, w, ho, wo, sp, pk) for e0, e1, cin, cout, h, w, ho, wo, sp, pk in calls), reverse=True)
print(f'{len(rows)} launches, {sum(r[0] for r in rows):.2f} ms')
for t, cin, cout, h, w, ho, wo, sp, pk in rows:
    fl = 2.0 * 16 * ho * wo * cin * cout * 9
    print(f'{t:7.3f} ms  {cin:4d} -> {cout:4d}  in {h:3d}x{w:3d} out {ho:3d}x{wo:3d}  subpix {sp} records {int(pk)}  {fl / t / 1e9:6.0f} TF')
