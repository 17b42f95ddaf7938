#!/bin/bash
# Tuning aid: which phase of mha_bwd_fused_kernel the time goes to.  Builds of attention.hip with -DMMNAS_DBG_MHA=<bits>
# (phases left out; wrong results) timed with tools/mha_bench.py: 1 = S^T / dA^T products, 2 = softmax backward VALU +
# dbias stores, 4 = dQ products, 8 = transposition + dK / dV products + LDS accumulation, 16 = the LDS accumulation only,
# 32 = the strided Q / dO operand loads of the prologue, 64 = the whole tile loop (prologue + epilogue remain).
L=mmnas_amd/lib
echo "variant 0 (the product kernel)"; python tools/mha_bench.py | grep "H=4 Sq=100 Sk=100"
for v in 1 2 3 4 8 16 32 64; do
  [ -f $L/libmmnas_hip_mha$v.so ] || continue
  echo "variant $v"; MMNAS_LIB_PATH=$PWD/$L/libmmnas_hip_mha$v.so python tools/mha_bench.py | grep "H=4 Sq=100 Sk=100"
done
