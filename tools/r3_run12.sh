#!/bin/bash
set -u
export TMPDIR=/tmp
show='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],3), round(1000/d["value_max"],3), round(1000/d["value_min"],3), "gemm", round(d["roofline"]["achieved"],1) if "roofline" in d else "")
'
for wl in search_vqa train_vqa; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "$show" ${wl}_default
  MMNAS_GEMM_TILE=12864 timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "$show" ${wl}_tile12864
  MMNAS_GEMM_PF=1 timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "$show" ${wl}_pf1
  MMNAS_GEMM_XCD=0 timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "$show" ${wl}_xcd0
  MMNAS_GEMM_SPLIT_P=48 timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "$show" ${wl}_splitp48
  MMNAS_GEMM_SPLIT_P=12 timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "$show" ${wl}_splitp12
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 --gemm-split 0 2>/dev/null | python3 -c "$show" ${wl}_fp32mfma
done
python tools/rel_bench.py 2>/dev/null | grep "B="
