#!/bin/bash
echo "=== gemm tests, two-role kernel with yield"; MMNAS_GEMM_SPEC=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k gemm 2>&1 | tail -2
echo "=== ksweep plain"; MMNAS_GEMM_SPEC=0 KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -2
for v in "" _y3 _y5; do echo "=== ksweep two-role, lib$v"; MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip$v.so MMNAS_GEMM_SPEC=1 KSWEEP_N=256 python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -2; done
