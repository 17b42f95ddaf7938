python -m pytest tests/test_kernels_gpu.py -x -q -k "mha or attention" 2>&1 | tail -4
python tools/mha_bench.py 2>/dev/null | head -4
