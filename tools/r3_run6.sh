#!/bin/bash
# round-3 GPU call 6: where the one-rank exchange's time goes: A/B of the row exchange / early scatter + a trace
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run6
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "embedding or pack or two_ranks" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
show='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],3), round(1000/d["value_max"],3), round(1000/d["value_min"],3), round(d["host_issue_ms_per_step"],3))
'
run() { name=$1; shift; env "$@" timeout 600 python bench.py --workload search_vqa_dp1 --no-cpu-baseline --no-prof --repeats 7 2>/dev/null | python3 -c "$show" $name; }
timeout 600 python bench.py --workload search_vqa --no-cpu-baseline --no-prof --repeats 7 2>/dev/null | python3 -c "$show" plain
run rows1_early1 MMNAS_DP_ROWS=1 MMNAS_DP_EARLY_SCATTER=1
run rows0_early1 MMNAS_DP_ROWS=0 MMNAS_DP_EARLY_SCATTER=1
run rows1_early0 MMNAS_DP_ROWS=1 MMNAS_DP_EARLY_SCATTER=0
run rows0_early0 MMNAS_DP_ROWS=0 MMNAS_DP_EARLY_SCATTER=0
W=/tmp/tr6; rm -rf $W; mkdir -p $W
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $W/a -o t -- python3 $ROOT/bench.py --workload search_vqa_dp1 --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-prof > $O/trace.log 2>&1)
python3 tools/step_timeline.py $W/a onehot_rows --list > $O/timeline_search_vqa_dp1.txt 2>&1
grep -n "pack_args\|@q3\|embedding\|CatArray\|rccl\|nccl" $O/timeline_search_vqa_dp1.txt | tail -40
