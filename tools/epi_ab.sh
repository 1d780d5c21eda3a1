#!/bin/bash
# usage: tools/epi_ab.sh libA.so libB.so ...  -- forward (record input) and data-gradient (fp32 input) layers of the one-wave
# kernels for several builds, alternating on one device (the epilogue A/Bs of round 6)
for r in 1 2; do for l in "$@"; do echo "== $l run $r"; EOVAE_LIB=$PWD/$l timeout -k 10 300 python tools/w4_bench.py 128,128,256 256,256,128 512,512,64 2>&1 | grep -v amdgpu.ids | sed 's/8-wave.*| w4 /w4 /' ; done; done
