#!/bin/bash
# A/B of the short-sequence backward (small.hip: sa_small_bwd_kernel, MMNAS_SMALL_BWD) on ONE box, alternating runs.
# bash tools/small_bwd_ab.sh > profiles/r05_small_bwd_ab.txt   (each line: median ms per step of 5 blocks of 20 steps)
one() {  # workload small_bwd
  MMNAS_SMALL_BWD=$2 python3 bench.py --workload $1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('%-16s small_bwd=%s  %.4f ms/step  (short-sequence ops %.3f gemm %.3f attention cores %.3f rowops %.3f ms; GEMM launches/step %.1f)' % ('$1', '$2', d['ms_per_step'], k.get('small_ops', 0), k.get('gemm', 0), k.get('mha_fwd', 0) + k.get('mha_bwd', 0), k.get('rowops', 0), d['roofline']['launches_per_step']))"
}
echo "# round 5: short-sequence backward A/B on one MI355X box, alternating runs of python3 bench.py --workload W (median of 5 blocks of 20 steps)"
for rep in 1 2 3; do
  for wl in search_vqa arch_vqa; do one $wl 1; one $wl 0; done
done
for wl in train_vqa search_vqa_unpad; do one $wl 1; one $wl 0; one $wl 1; one $wl 0; done
