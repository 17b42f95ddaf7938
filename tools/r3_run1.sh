#!/bin/bash
# round-3 GPU call 1: GEMM parity, A/B against the round-2 build, LDS counters per layout, short bench
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run1
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/test_kernels.log 2>&1; echo "kernels rc=$?" >> $O/test_kernels.log
timeout 600 python tools/gemm_ab.py mmnas_amd/lib/libmmnas_hip_r2.so mmnas_amd/lib/libmmnas_hip.so > $O/gemm_ab.txt 2>&1
W=/tmp/pmc1; rm -rf $W; mkdir -p $W
(cd /tmp && GEMM_PMC_LAYOUTS=1 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $W/lds -o t -- python3 $ROOT/tools/gemm_pmc.py > $O/pmc_lds.log 2>&1)
(cd /tmp && GEMM_PMC_LAYOUTS=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $W/mfma -o t -- python3 $ROOT/tools/gemm_pmc.py > $O/pmc_mfma.log 2>&1)
python3 tools/pmc_counters.py $O/pmc_gemm_layouts.json $W/lds $W/mfma > $O/pmc_summary.txt 2>&1
timeout 600 python bench.py --workload search_vqa --no-cpu-baseline > $O/bench_search.json 2> $O/bench_search.err
timeout 600 python bench.py --workload train_vqa --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err
tail -3 $O/test_kernels.log; cat $O/pmc_summary.txt; cut -c1-400 $O/bench_search.json; cut -c1-300 $O/bench_train.json
