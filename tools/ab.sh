#!/bin/bash
# usage: tools/ab.sh libA.so libB.so  -- interleaved layer_bench runs on one device
for i in 1 2; do for l in "$@"; do echo "== $l"; EOVAE_LIB=$l python tools/layer_bench.py bf16x3 2>&1 | tail -6 | cut -c1-110; done; done
