"""Build the C-ABI HIP library (libeovae_hip.so) in-tree with hipcc for gfx950.

    python -m eo_vae_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libeovae_hip.so')
ARCH = 'gfx950'


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h'))
    return any(os.path.getmtime(s) > t for s in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libeovae_hip.so')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, 'build', os.path.basename(src) + '.o')
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(h)
                        for h in glob.glob(os.path.join(CSRC, '*.h')))):
            continue
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj]
        if os.environ.get('EOVAE_BUILD_PARKED') == '1':    # the measured-and-parked kernels too (csrc/eovae_common.h); use --force
            cmd.insert(1, '-DEOVAE_PARKED=1')
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out.decode()}')
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
