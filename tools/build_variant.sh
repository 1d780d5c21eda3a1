#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG=..]...   -> eo_vae_amd/build/exp/lib_NAME.so
# Rebuilds one kernel source (SRC=conv_f16c8; default conv_bf16x3) with extra flags (timing-only ablations / experiments) and links it with the
# other objects of the last `python -m eo_vae_amd.build`.  The exp/ directory travels to the GPU box
# (gpurun_out/ does not); delete it afterwards.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=${SRC:-conv_bf16x3}
mkdir -p eo_vae_amd/build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c eo_vae_amd/csrc/$src.hip -o eo_vae_amd/build/exp/conv_$name.o
objs=$(ls eo_vae_amd/build/*.o | grep -v "/$src.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eo_vae_amd/build/exp/lib_$name.so $objs eo_vae_amd/build/exp/conv_$name.o
rm eo_vae_amd/build/exp/conv_$name.o
echo eo_vae_amd/build/exp/lib_$name.so
