#!/bin/bash
# round-3 GPU call 4: timeline of the one-rank RCCL exchange (search_vqa_dp1 / train_vqa_dp1) and of the plain steps
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run4
mkdir -p $O
W=/tmp/tr4; rm -rf $W; mkdir -p $W
for wl in search_vqa_dp1 train_vqa_dp1; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $W/$wl -o t -- python3 $ROOT/bench.py --workload $wl --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-prof > $O/trace_$wl.log 2>&1)
  marker=onehot_rows; [ $wl = train_vqa_dp1 ] && marker=row_is_zero
  python3 tools/step_timeline.py $W/$wl $marker --list > $O/timeline_$wl.txt 2>&1
  head -1 $(find $W/$wl -name '*kernel_trace.csv' | head -1) > $O/csv_header_$wl.txt
done
head -45 $O/timeline_search_vqa_dp1.txt
