mkdir -p gpurun_out/r6d
./tools/probe/tr_read_probe | head -20
python -m pytest tests/test_kernels_gpu.py -x -q -k "mha or attention" 2>&1 | tail -15
python tools/mha_bench.py 2>/dev/null | tee gpurun_out/r6d/mha_bench.txt
MMNAS_MHA_BWD_B16=0 python tools/mha_bench.py 2>/dev/null | head -2
