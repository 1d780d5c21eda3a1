"""Which gradient tensors still need a separate max|t| pass (no producer tag)?  One train step, shapes + callers."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eo_vae_amd import ops
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
wl = WAVELENGTHS['S2RGB']
model = bench.build_model(3).cuda().train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
b = {'image': synthetic_images(16, 3, 256, g).cuda(), 'wvs': torch.tensor(wl).cuda(), 'modality': 'S2RGB'}
drv.step(b, 0)
seen = collections.Counter()
orig = ops.absmax_bits
def spy(t):
    fr = [f for f in traceback.extract_stack()[:-1] if 'ops.py' in f.filename or 'layers.py' in f.filename or 'model.py' in f.filename]
    seen[(tuple(t.shape), ' < '.join(f'{f.name}:{f.lineno}' for f in fr[-4:]))] += 1
    return orig(t)
ops.absmax_bits = spy
drv.step(b, 1)
torch.cuda.synchronize()
for (shape, where), n in sorted(seen.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3] if len(kv[0][0]) == 4 else 0):
    print(n, shape, where)
