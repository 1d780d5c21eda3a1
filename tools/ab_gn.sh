set -e
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels2.py -x -q -m gpu > gpurun_out/t_wg.log 2>&1 || true
tail -5 gpurun_out/t_wg.log
python tools/layer_bench.py f16c8 2>&1 | grep f16c8 > gpurun_out/lb_wg.log
python bench.py --no-cpu-baseline > gpurun_out/bench_wg.json 2>gpurun_out/bench_wg.err
