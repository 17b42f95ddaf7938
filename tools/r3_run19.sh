#!/bin/bash
# decomposition of the K loop's per-tile cost: default / one tile in flight / 3 products / fp32 MFMA / no conversion / no MFMA / neither
export KSWEEP_N=256
run() { echo "== $1"; shift; env "$@" python tools/gemm_ksweep.py 2>&1 | grep -v amdgpu.ids | head -3; }
run default X=0
run "one K-tile of loads in flight (MMNAS_GEMM_PF=1)" MMNAS_GEMM_PF=1
run "3 bf16 products (MMNAS_GEMM_SPLIT=3)" MMNAS_GEMM_SPLIT=3
run "fp32 MFMA (MMNAS_GEMM_SPLIT=0)" MMNAS_GEMM_SPLIT=0
run "conversion replaced by a bit copy (-DMMNAS_DBG_NOCONV)" MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_nc.so
run "no fragment reads / MFMAs (-DMMNAS_DBG_NOMFMA)" MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_nm.so
run "neither: loads, bit copy to LDS, barriers" MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_ncm.so
