#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
echo "=== A (packed subtractions) vs B (scalar subtractions)"
python tools/gemm_ab.py mmnas_amd/lib/libmmnas_hip_a.so mmnas_amd/lib/libmmnas_hip.so 2>&1 | grep -v amdgpu.ids
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa train_vqa; do for v in _a ""; do
  echo "== $wl lib$v"
  MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip$v.so $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
done; done; done
