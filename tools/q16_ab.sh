#!/bin/bash
# usage: tools/q16_ab.sh libA.so libB.so ...  -- the MX forward layers through conv_f16c8_q16.hip of several builds, alternating on one device
for r in 1 2; do for l in "$@"; do echo "== $l run $r"; EOVAE_LIB=$PWD/$l timeout -k 10 300 python tools/w4_bench.py 128,128,256 256,256,128 512,512,64 256,128,256 512,256,128 2>&1 | grep -v amdgpu.ids | grep -v dgrad | sed 's/.*| q16/q16/' ; done; done
