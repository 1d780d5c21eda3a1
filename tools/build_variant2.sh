#!/bin/bash
# usage: tools/build_variant2.sh SRC NAME [-DFLAG=..]...  -> eo_vae_amd/build/exp/lib_NAME.so with csrc/SRC rebuilt with extra flags
set -e
cd "$(dirname "$0")/.."
src=$1; name=$2; shift; shift
mkdir -p eo_vae_amd/build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c eo_vae_amd/csrc/$src -o eo_vae_amd/build/exp/v_$name.o
objs=$(ls eo_vae_amd/build/*.o | grep -v "/${src}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eo_vae_amd/build/exp/lib_$name.so $objs eo_vae_amd/build/exp/v_$name.o
rm eo_vae_amd/build/exp/v_$name.o
echo eo_vae_amd/build/exp/lib_$name.so
