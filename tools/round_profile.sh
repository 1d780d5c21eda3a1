# usage (on the GPU box): bash tools/round_profile.sh TAG
# bench JSON line, rocprofv3 kernel-trace stats of the same command, and the two PMC traffic passes;
# everything lands in gpurun_out/ (copy what is to be judged into profiles/).
R=$GRAFT_REPO_ROOT
cd $R && python3 bench.py --steps 10 --warmup 3 > gpurun_out/$1_bench_cfg2.json 2> gpurun_out/$1_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$1_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$1_stats.log 2>&1
cp $(find $R/gpurun_out/$1_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$1_bench_cfg2_kernel_stats.csv
tail -1 $R/gpurun_out/$1_stats.log | cut -c1-300 > /dev/null
bash $R/tools/pmc_traffic.sh $1 > /dev/null 2>&1
cat $R/gpurun_out/$1_bench_cfg2.json
