timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels2.py -x -q -m gpu > gpurun_out/t_wg.log 2>&1; tail -3 gpurun_out/t_wg.log
for l in eo_vae_amd/build/exp/lib_base.so eo_vae_amd/libeovae_hip.so eo_vae_amd/build/exp/lib_base.so eo_vae_amd/libeovae_hip.so; do echo "== $l"; LB_XPACK=1 LB_RES=1 LB_GN=1 EOVAE_LIB=$l python tools/layer_bench.py f16c8 2>&1 | grep f16c8 | cut -c1-52; done
for l in eo_vae_amd/build/exp/lib_base.so eo_vae_amd/libeovae_hip.so; do echo "== $l plain"; EOVAE_LIB=$l python tools/layer_bench.py f16c8 2>&1 | grep f16c8 | cut -c1-52; done
