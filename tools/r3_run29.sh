#!/bin/bash
echo "=== gemm tests, two-role kernel"; MMNAS_GEMM_SPEC=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k gemm 2>&1 | tail -3
for sp in 0 1; do echo "=== ksweep MMNAS_GEMM_SPEC=$sp"; MMNAS_GEMM_SPEC=$sp KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -5; done
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa train_vqa; do for sp in 0 1; do
  echo "== $wl MMNAS_GEMM_SPEC=$sp"
  MMNAS_GEMM_SPEC=$sp $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
done; done; done
