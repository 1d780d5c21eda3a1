"""Does the step time depend on WHERE the allocator's blocks sit?  One process, one model; between trials the caching
allocator's blocks are returned to the driver (torch.cuda.empty_cache()), so the next steps run on freshly mapped memory.
    python tools/placement_probe.py [trials]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eo_vae_amd.data import WAVELENGTHS, synthetic_images
from eo_vae_amd.dynamic_conv import prefer_rocblas_for_generator

prefer_rocblas_for_generator()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda', 0)
if os.environ.get('PROBE_PREALLOC'):
    _big = torch.empty(int(os.environ['PROBE_PREALLOC']) << 30, dtype=torch.uint8, device=dev)   # ONE driver allocation that the
    del _big                                                                                      # caching allocator then carves up
wl = WAVELENGTHS['S2RGB']
model = bench.build_model(len(wl)).to(dev).train()
drv = bench.StepDriver(model, 1)
g = torch.Generator('cpu').manual_seed(1234)
batch = {'image': synthetic_images(16, len(wl), 256, g).to(dev), 'wvs': torch.tensor(wl, device=dev), 'modality': 'S2RGB'}
k = 0
from eo_vae_amd import ops
for t in range(trials):
    if os.environ.get('PROBE_TIMERS') == '1' or (os.environ.get('PROBE_TIMERS') == 'alt' and t % 2 == 1):
        ops.CONV3X3_TIMER = ops.KernelTimer(); ops.WGRAD3X3_TIMER = ops.KernelTimer()
    else:
        ops.CONV3X3_TIMER = None; ops.WGRAD3X3_TIMER = None
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    if len(sys.argv) > 2:
        time.sleep(float(sys.argv[2]))
    for _ in range(3):
        drv.step(batch, k); k += 1
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ev[0].record()
    for i in range(8):
        drv.step(batch, k); k += 1
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(8)]
    print(f'trial {t}: ' + ' '.join(f'{m:6.1f}' for m in ms) + f' | reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB', flush=True)
