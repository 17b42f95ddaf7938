#!/bin/bash
# round-3 GPU call 2: transposing-store lane maps A/B (TN / NN products), parity of the e-flip map, bench with it
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run2
mkdir -p $O
L=mmnas_amd/lib
for v in 0 1 2; do
  GEMM_AB_ONLY=TN,NN timeout 600 python tools/gemm_ab.py $L/libmmnas_hip_r2.so $L/libmmnas_hip_tmap$v.so > $O/gemm_ab_tmap$v.txt 2>&1
done
MMNAS_LIB_PATH=$ROOT/$L/libmmnas_hip_tmap2.so timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or products" > $O/test_kernels_tmap2.log 2>&1
for v in 0 1 2; do
  MMNAS_LIB_PATH=$ROOT/$L/libmmnas_hip_tmap$v.so timeout 600 python bench.py --workload search_vqa --no-cpu-baseline --no-prof 2>/dev/null | cut -c1-330 > $O/bench_search_tmap$v.json
  MMNAS_LIB_PATH=$ROOT/$L/libmmnas_hip_tmap$v.so timeout 600 python bench.py --workload train_vqa --no-cpu-baseline --no-prof 2>/dev/null | cut -c1-330 > $O/bench_train_tmap$v.json
done
MMNAS_LIB_PATH=$ROOT/$L/libmmnas_hip_r2.so timeout 600 python bench.py --workload search_vqa --no-cpu-baseline --no-prof 2>/dev/null | cut -c1-330 > $O/bench_search_r2.json
MMNAS_LIB_PATH=$ROOT/$L/libmmnas_hip_r2.so timeout 600 python bench.py --workload train_vqa --no-cpu-baseline --no-prof 2>/dev/null | cut -c1-330 > $O/bench_train_r2.json
tail -2 $O/test_kernels_tmap2.log
grep -h "^TN" $O/gemm_ab_tmap*.txt | head -60
for f in $O/bench_*.json; do echo $f; cut -c1-200 $f; done
