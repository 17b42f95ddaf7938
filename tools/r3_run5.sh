#!/bin/bash
# round-3 GPU call 5: exchange changes (row exchange, balanced pack, early scatter): tests + dp1 records
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run5
mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py tests/test_dropin_gpu.py tests/test_harness_gpu.py -x -q -m gpu -k "embedding or pack or two_ranks or rccl or ddp or itm or optim or flat" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
for wl in search_vqa search_vqa_dp1 train_vqa train_vqa_dp1; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-prof 2>$O/bench_$wl.err | cut -c1-420 > $O/bench_$wl.json
  cut -c1-230 $O/bench_$wl.json
done
