#!/bin/bash
# A/B inside one call: encoder/decoder overlap, side stream, head overlap vs the default single-stream step
B="python bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-prof"
for rep in 1 2; do
for wl in search_vqa train_vqa; do
  for env in "X=0" "MMNAS_CHAIN_OVERLAP=1" "MMNAS_SIDE_STREAM=1" "MMNAS_HEAD_OVERLAP=1"; do
    echo "== $wl $env"
    env $env $B --workload $wl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_min'], d['value_max'])"
  done
done
done
