#!/bin/bash
# Regenerates everything under profiles/ for one round on a GPU box (run from the repo root):
#   bash tools/refresh_profiles.sh r01
# bench lines (with the CPU baseline for the headline workload), rocprofv3 --kernel-trace --stats summaries of the
# same command, and the two --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass) behind roofline.traffic.
# Raw traces stay in /tmp; only the summaries are written to profiles/.
set -u
R=${1:-r01}
ROOT=$PWD
export TMPDIR=/tmp
W=/tmp/mmnas_prof
rm -rf $W; mkdir -p $W profiles
for wl in train_vqa search_vqa; do
  cmd="bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace_$wl -o t -- python3 $ROOT/$cmd > $W/trace_$wl.log 2>&1)
  python3 tools/summarize_prof.py $W/trace_$wl profiles/${R}_$wl 23 \
    "rocprofv3 --kernel-trace --stats -- python3 $cmd  (3 warm-up + 10 timed + 10 roofline-pass steps)"
  [ -n "${SKIP_PMC:-}" ] && continue
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $W/pmc_${wl}_$c -o t -- python3 $ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $W/pmc_${wl}_$c.log 2>&1)
  done
  python3 tools/pmc_traffic.py $W/pmc_${wl}_FETCH_SIZE $W/pmc_${wl}_WRITE_SIZE profiles/${R}_traffic_$wl.json
done
# bench lines last: they read the traffic files written above
python3 bench.py --workload train_vqa | grep '^{' | tail -1 > profiles/${R}_bench_train_vqa.json
python3 bench.py --workload search_vqa --no-cpu-baseline | grep '^{' | tail -1 > profiles/${R}_bench_search_vqa.json
ls -la profiles/
for f in $W/*.log; do echo "== $f"; tail -n 3 $f; done
