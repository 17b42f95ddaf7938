mkdir -p gpurun_out/r6a
python -m pytest tests/test_gemm_ln_gpu.py -x -q 2>&1 | tail -8
python tools/gemm_ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6a/gemm_ln_bench7.txt
