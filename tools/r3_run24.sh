#!/bin/bash
for t in 16 8 4; do echo "=== MMNAS_GEMM_HYB_T=$t"; MMNAS_GEMM_HYB_T=$t python tools/gemm_ab.py mmnas_amd/lib/libmmnas_hip.so mmnas_amd/lib/libmmnas_hip.so 2>&1 | grep "N=256" | grep "6400" | grep -v amdgpu; done
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa arch_vqa; do for t in 16 8; do
  echo "== $wl HYB_T=$t"
  MMNAS_GEMM_HYB_T=$t $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
done; done; done
