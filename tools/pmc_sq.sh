# usage (on the GPU box): bash tools/pmc_sq.sh LIBNAME  -> gpurun_out/pmcsq_LIBNAME/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export EOVAE_LIB=$R/eo_vae_amd/build/exp/lib_$1.so
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq_$1 -- python3 $R/tools/pmc_layers.py > $R/gpurun_out/pmcsq_$1.log 2>&1 || echo "pass failed"
