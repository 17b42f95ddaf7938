#!/bin/bash
# A/B of the two chain hoists of round 5 on ONE box, alternating runs (profiles/r05_hoist_ab.txt):
#   MMNAS_REL_HOIST    the relation bias of all relation operators of a stream in one launch per direction (relmulti.hip)
#   MMNAS_GUIDED_HOIST the key / value projections of all guided operators as grouped launches
# bash tools/hoist_ab.sh > profiles/r05_hoist_ab.txt      (each line: median ms per step of 5 blocks of 20 steps)
one() {  # workload rel guided
  MMNAS_REL_HOIST=$2 MMNAS_GUIDED_HOIST=$3 python3 bench.py --workload $1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('%-12s rel_hoist=%s guided_hoist=%s  %.4f ms/step  (rel_fwd %.3f rel_bwd %.3f gemm %.3f ms; GEMM launches/step %.1f)' % ('$1', '$2', '$3', d['ms_per_step'], k.get('rel_fwd', 0), k.get('rel_bwd', 0), k.get('gemm', 0), d['roofline']['launches_per_step']))"
}
echo "# round 5: chain hoists A/B on one MI355X box, alternating runs of python3 bench.py --workload W (median of 5 blocks of 20 steps)"
for rep in 1 2 3; do
  for wl in search_vqa arch_vqa; do
    one $wl 1 1; one $wl 0 1; one $wl 1 0; one $wl 0 0
  done
done
for wl in train_vqa search_vqa_unpad; do one $wl 1 1; one $wl 0 0; one $wl 1 1; one $wl 0 0; done
