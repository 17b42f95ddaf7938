#!/bin/bash
echo "=== A (ILV=0) vs C (ILV=1, PRE=5, occupancy 3)"
python tools/gemm_ab.py mmnas_amd/lib/libmmnas_hip_a.so mmnas_amd/lib/libmmnas_hip_c.so 2>&1 | grep -v amdgpu.ids
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa train_vqa; do for v in _a _c _d; do
  echo "== $wl lib$v"
  MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip$v.so $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
done; done; done
