# usage (on the GPU box): bash tools/pmc_traffic.sh TAG
# Two SEPARATE passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip traces)
# over the bench workload, then tools/pmc_traffic.py TAG folds them into gpurun_out/TAG_pmc_hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/$1_pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$1_pmc_$c.log 2>&1 || echo "$c pass failed"
done
python3 $R/tools/pmc_traffic.py $1
