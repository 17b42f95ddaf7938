R=$PWD
for v in pf2 pf4 pf8; do
  echo "== $v"; MMNAS_LIB_PATH=$R/mmnas_amd/lib/libmmnas_hip_$v.so python tools/gemm_ln_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/gemm_ln_bench8.txt
