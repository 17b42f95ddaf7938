mkdir -p gpurun_out/r6c
MMNAS_TEST_REL_SELF_TOL=3e-4 python -m pytest tests/test_chain_gpu.py tests/test_configs_gpu.py tests/test_dropin_gpu.py tests/test_harness_gpu.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r6c/rel_self_tol_3e-4.log
MMNAS_COMMIT=$1 bash tools/refresh_profiles.sh r06 > gpurun_out/r6c/refresh.log 2>&1
tail -5 gpurun_out/r6c/refresh.log
