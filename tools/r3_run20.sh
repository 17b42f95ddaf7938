#!/bin/bash
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do for wl in search_vqa train_vqa arch_vqa; do for v in _a ""; do
  echo "== $wl lib$v"
  MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip$v.so $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
done; done; done
