"""Host-side time of one train step (enqueue only, no synchronisation) against its GPU time: is the step
launch-bound?  python tools/host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
wl = WAVELENGTHS['S2RGB']
model = bench.build_model(3).cuda().train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
b = {'image': synthetic_images(16, 3, 256, g).cuda(), 'wvs': torch.tensor(wl).cuda(), 'modality': 'S2RGB'}
for i in range(3):
    drv.step(b, i)
torch.cuda.synchronize()
host, wall = [], []
for i in range(6):
    t0 = time.perf_counter()
    drv.step(b, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    wall.append((t2 - t0) * 1e3)
print('host enqueue ms per step:', [round(h, 1) for h in host])
print('wall (enqueue + drain) ms :', [round(w, 1) for w in wall])
