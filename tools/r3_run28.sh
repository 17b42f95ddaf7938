#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k gemm 2>&1 | tail -2
echo "=== ksweep, default lib (s_nop 1 behind every MFMA)"; KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -3
echo "=== ksweep, no yield"; MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_a.so KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -3
echo "=== ksweep, s_nop 0"; MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_y0.so KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -3
echo "=== ksweep, s_nop 3"; MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_y3.so KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -3
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa train_vqa arch_vqa; do for v in _a "" _y0 _y3; do
  echo "== $wl lib$v"
  MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip$v.so $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
done; done; done
