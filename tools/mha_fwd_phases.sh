#!/bin/bash
# Tuning aid: which phase of mha_fwd_kernel costs what.  Builds of attention.hip with one phase of the forward left out
# (-DMMNAS_DBG_FWD=<mask>: 1 S^T products, 2 softmax arithmetic, 4 P V products, 8 output stores; results are WRONG by
# design) are timed with tools/mha_bench.py; the libraries are mmnas_amd/lib/libmmnas_hip_fwd<mask>.so (built on the host:
#   for D in 1 2 4 8 7 15; do hipcc ... -DMMNAS_DBG_FWD=$D -c attention.hip -o build_fwd$D/attention.o; hipcc -shared ... ; done)
R=$PWD
echo "mask 0 (the product kernel)"; python3 tools/mha_bench.py 2>/dev/null | head -2
for D in 1 2 4 8 7 15; do
  [ -f mmnas_amd/lib/libmmnas_hip_fwd$D.so ] || continue
  echo "mask $D left out"; MMNAS_LIB_PATH=$R/mmnas_amd/lib/libmmnas_hip_fwd$D.so python3 tools/mha_bench.py 2>/dev/null | head -2
done
