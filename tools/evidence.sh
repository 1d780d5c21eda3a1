# usage (on the GPU box): bash tools/evidence.sh TAG WORKLOAD [sq]
# rocprofv3 evidence of `python3 bench.py --workload WORKLOAD`: kernel-trace stats, the two SEPARATE HBM-traffic passes
# (FETCH_SIZE, WRITE_SIZE) and -- with a third argument -- two SQ passes (MFMA busy / issue, LDS conflicts).  --pmc is only ever
# combined with --kernel-trace; the program stands directly behind `--`.  tools/evidence_fold.py folds everything per kernel
# into gpurun_out/TAG_WORKLOAD_kernels.json (time per step, HBM bytes and GB/s per launch, SQ shares).
R=$GRAFT_REPO_ROOT
T=$1; W=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_${W}_stats -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${T}_${W}_stats.log 2>&1 || echo "stats pass failed"
cp $(find $R/gpurun_out/${T}_${W}_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${T}_bench_${W}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${T}_${W}_pmc_$c -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_${W}_pmc_$c.log 2>&1 || echo "$c pass failed"
done
if [ -n "$3" ]; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/${T}_${W}_pmc_SQA -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_${W}_pmc_SQA.log 2>&1 || echo "SQ pass A failed"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_${W}_pmc_SQB -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_${W}_pmc_SQB.log 2>&1 || echo "SQ pass B failed"
fi
python3 $R/tools/evidence_fold.py $T $W
