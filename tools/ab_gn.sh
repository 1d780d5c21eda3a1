set -e
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "epilogue_groupnorm or operand_records or f16c8" > gpurun_out/t_gn.log 2>&1
for g in 0 1 0 1; do echo "== LB_GN=$g"; LB_RES=1 LB_GN=$g python tools/layer_bench.py f16c8 2>&1 | grep f16c8 | cut -c1-50; done > gpurun_out/lb_gn.log 2>&1
for t in 0 1 0b 1b; do EOVAE_GN_FUSED=${t:0:1} python bench.py --no-cpu-baseline > gpurun_out/bench_gn$t.json 2>gpurun_out/bench_gn$t.err; done
