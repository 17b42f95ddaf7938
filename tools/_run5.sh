mkdir -p gpurun_out/r6a
bash tools/mha_fwd_phases.sh 2>&1 | tee gpurun_out/r6a/mha_fwd_phases.txt
python -m pytest tests/test_gemm_ln_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r6a/pytest_gemm_ln.log
python tools/gemm_ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6a/gemm_ln_bench.txt
