#!/bin/bash
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa train_vqa arch_vqa; do for kv in 0 1; do
  echo "== $wl HIP_FORCE_DEV_KERNARG=$kv"
  HIP_FORCE_DEV_KERNARG=$kv $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'], d.get('host_issue_ms_per_step_empty_queue'))"
done; done; done
for kv in 0 1; do echo "== ksweep HIP_FORCE_DEV_KERNARG=$kv"; HIP_FORCE_DEV_KERNARG=$kv KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -3; done
