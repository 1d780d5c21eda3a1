# usage (on the GPU box): bash tools/pmc_sq4.sh TAG -> gpurun_out/pmcsq3{a,b}_TAG_wgrad_bench/ + gpurun_out/TAG_pmc_sq.json  (round 4)
# SQ counters of the two plain weight-gradient kernels (tile kernel, row-streaming kernel) over tools/wgrad_bench.py.
# One hardware block per pass, no trace domains besides --kernel-trace.  Quad-cycle counters except SQ_VALU_MFMA_BUSY_CYCLES.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prog=wgrad_bench
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq3a_$1_$prog -- python3 $R/tools/$prog.py > $R/gpurun_out/pmcsq3a_$1_$prog.log 2>&1 || echo "pass A failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq3b_$1_$prog -- python3 $R/tools/$prog.py > $R/gpurun_out/pmcsq3b_$1_$prog.log 2>&1 || echo "pass B failed"
python3 $R/tools/pmc_sq3_fold.py $1
