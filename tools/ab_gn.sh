cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02h_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02h_stats.log 2>&1
cp $(find $R/gpurun_out/r02h_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02h_kernel_stats.csv
