"""Which host ops own the library (Tensile `Cijk_*`) GEMMs and the ATen reductions / elementwise kernels of one
train step?  torch.profiler over one step of the bench workload, grouped by (op, input shapes).
python tools/op_attrib.py > gpurun_out/op_attrib.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    drv.step(b, 3)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=70,
                                                         max_name_column_width=60, max_shapes_column_width=90))
