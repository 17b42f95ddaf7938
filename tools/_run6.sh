mkdir -p gpurun_out/r6a
R=$PWD
python -m pytest tests/test_gemm_ln_gpu.py tests/test_kernels_gpu.py -x -q -k "gemm_ln or mha or attention" 2>&1 | tail -3
python tools/mha_bench.py 2>/dev/null | tee gpurun_out/r6a/mha_bench6.txt
for v in "" _ln1 _ln2 _ln3 _ln4 _ln5; do
  echo "== lib$v"; MMNAS_LIB_PATH=$R/mmnas_amd/lib/libmmnas_hip$v.so python tools/gemm_ln_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6a/gemm_ln_bench6.txt
