#!/bin/bash
# Tuning aid: which phase of mha_bwd_b16_kernel (attention_bwd16.hip) costs what.  Builds with one phase left out
# (-DB16_DBG=<mask>: 1 S^T / dA^T products, 2 softmax backward, 4 dQ products + accumulation, 8 transposition + dK^T / dV^T
# products, 16 every step: prologue + epilogue only; results are WRONG by design) timed with tools/mha_bench.py;
# libraries mmnas_amd/lib/libmmnas_hip_b<mask>.so (built on the host, see the loop in docs/LAB_NOTES.md round 6).
R=$PWD
echo "mask 0 (the product kernel)"; python3 tools/mha_bench.py 2>/dev/null | head -2
for D in 1 2 4 8 3 12 16; do
  [ -f mmnas_amd/lib/libmmnas_hip_b$D.so ] || continue
  echo "mask $D left out"; MMNAS_LIB_PATH=$R/mmnas_amd/lib/libmmnas_hip_b$D.so python3 tools/mha_bench.py 2>/dev/null | head -2
done
