python -m pytest tests/test_kernels_gpu.py tests/test_small_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -4
python tools/mha_bench.py 2>/dev/null
MMNAS_MHA_FWD_B16=0 python tools/mha_bench.py 2>/dev/null | head -2
