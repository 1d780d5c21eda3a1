timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1; tail -3 gpurun_out/t_full.log
for t in 0 1 0b 1b; do EOVAE_ZERO_CHUNKS=${t:0:1} python bench.py --no-cpu-baseline > gpurun_out/bench_z$t.json 2>gpurun_out/bench_z$t.err; done
