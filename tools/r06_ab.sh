#!/bin/bash
# A/B of round 6's switches on ONE box, alternating runs of python3 bench.py --workload W (median of 5 blocks of 20 steps):
#   MMNAS_GEMM_LN       merge projection + dropout + residual + LayerNorm as one row-panel launch (gemmln.hip; d = 256, K <= 256)
#   MMNAS_DP_TAIL_MAIN  end of the data-parallel step on the backward's own stream + one scatter launch (dp.py)
# bash tools/r06_ab.sh > profiles/r06_ab.txt
one() {  # workload ENV=VALUE
  env $2 python3 bench.py --workload $1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
r = d.get('roofline', {})
print('%-16s %-22s %.4f ms/step  (gemm %.3f rowops %.3f mha_fwd %.3f mha_bwd %.3f ms; GEMM launches/step %s)' % ('$1', '$2', d['ms_per_step'], k.get('gemm', 0), k.get('rowops', 0), k.get('mha_fwd', 0), k.get('mha_bwd', 0), r.get('launches_per_step')))"
}
echo "# round 6: switches A/B on one MI355X box, alternating runs (median of 5 blocks of 20 steps)"
for rep in 1 2 3; do
  for wl in search_vqa search_vqa_unpad; do one $wl MMNAS_GEMM_LN=1; one $wl MMNAS_GEMM_LN=0; done
done
for rep in 1 2 3; do one search_vqa_dp1 MMNAS_DP_TAIL_MAIN=1; one search_vqa_dp1 MMNAS_DP_TAIL_MAIN=0; done
for rep in 1 2; do one search_vqa MMNAS_DP_TAIL_MAIN=1; done
