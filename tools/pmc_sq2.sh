# usage (on the GPU box): bash tools/pmc_sq2.sh TAG -> gpurun_out/pmcsq2_TAG.csv
# SQ counters of the f16c8 convolution / weight-gradient kernels over tools/layer_bench.py (validated against
# `rocprofv3 -L`; one pass, no trace domains besides --kernel-trace).  Quad-cycle counters except MFMA_BUSY.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq2_$1 -- python3 $R/tools/layer_bench.py f16c8 > $R/gpurun_out/pmcsq2_$1.log 2>&1 || echo "pass A failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq2b_$1 -- python3 $R/tools/layer_bench.py f16c8 > $R/gpurun_out/pmcsq2b_$1.log 2>&1 || echo "pass B failed"
