#!/bin/bash
# round-3 GPU call 8: tests again + arch step kernel classes with / without the mixed chain + dp1 A/B
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run8
mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py tests/test_harness_gpu.py -q -m gpu -k "embedding or pack or two_ranks or harness" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log
show='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],3), round(1000/d["value_max"],3), round(1000/d["value_min"],3), round(d["host_issue_ms_per_step"],3))
        kc=d.get("kernel_classes")
        if kc: print("   ", {k:(round(v["ms_per_step"],3), round(v["launches_per_step"],1)) for k,v in kc.items()}, "kernel ms", round(d["roofline_pass"]["kernel_ms_per_step"],3))
'
for c in 0 1; do
  MMNAS_MIXED_CHAIN=$c timeout 600 python bench.py --workload arch_vqa --no-cpu-baseline 2>/dev/null | python3 -c "$show" arch_chain$c
done
timeout 600 python bench.py --workload search_vqa --no-cpu-baseline --no-prof --repeats 7 2>/dev/null | python3 -c "$show" plain
for r in 1 0; do for e in 1 0; do
  MMNAS_DP_ROWS=$r MMNAS_DP_EARLY_SCATTER=$e timeout 600 python bench.py --workload search_vqa_dp1 --no-cpu-baseline --no-prof --repeats 7 2>/dev/null | python3 -c "$show" dp1_rows${r}_early$e
done; done
timeout 600 python bench.py --workload train_vqa --no-cpu-baseline --no-prof --repeats 5 2>/dev/null | python3 -c "$show" train_plain
for r in 1 0; do
  MMNAS_DP_ROWS=$r timeout 600 python bench.py --workload train_vqa_dp1 --no-cpu-baseline --no-prof --repeats 5 2>/dev/null | python3 -c "$show" train_dp1_rows$r
done
