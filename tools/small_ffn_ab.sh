#!/bin/bash
# A/B of the one-launch FeedForward forward for short row counts (small.hip: ffn_small_fwd_kernel, MMNAS_SMALL_FFN) on ONE box.
# bash tools/small_ffn_ab.sh > profiles/r05_small_ffn_ab.txt   (each line: median ms per step of 5 blocks of 20 steps)
one() {  # workload small_ffn
  MMNAS_SMALL_FFN=$2 python3 bench.py --workload $1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('%-16s small_ffn=%s  %.4f ms/step  (short-sequence ops %.3f gemm %.3f rowops %.3f ms; GEMM launches/step %.1f)' % ('$1', '$2', d['ms_per_step'], k.get('small_ops', 0), k.get('gemm', 0), k.get('rowops', 0), d['roofline']['launches_per_step']))"
}
echo "# round 5: one-launch FeedForward forward (M = 896 rows) A/B on one MI355X box, alternating runs of python3 bench.py --workload W (median of 5 blocks of 20 steps)"
# small_ffn: 0 = the general path (two products + LayerNorm), 1 = four hidden slices of 256 per row group, 2 = eight of 128
for rep in 1 2 3; do
  for wl in search_vqa search_vqa_unpad; do one $wl 1; one $wl 2; one $wl 0; done
done
for wl in bilevel_vqa; do one $wl 1; one $wl 2; one $wl 0; done
