#!/bin/bash
# usage: tools/ab_c8.sh libA.so libB.so ...  -- f16c8 layer timings of several builds, interleaved on one device
for l in "$@"; do echo "== $l"; EOVAE_LIB=$l python tools/layer_bench.py f16c8 2>&1 | grep f16c8 | cut -c1-70; done
