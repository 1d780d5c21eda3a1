# usage (on the GPU box): bash tools/pmc_clk.sh LIBNAME  -> gpurun_out/pmcclk_LIBNAME/ (effective shader clock per dispatch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export EOVAE_LIB=$R/eo_vae_amd/build/exp/lib_$1.so
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcclk_$1 -- python3 $R/tools/pmc_layers.py > $R/gpurun_out/pmcclk_$1.log 2>&1 || echo "pass failed"
