#!/bin/bash
# Regenerates profiles/<round>_gemm_lean_ab.txt on a GPU box (run from the repo root; needs the stamp build:
#   make -C mmnas_amd/csrc variant NAME=stamp DEFS=-DMMNAS_DBG_STAMP=37   -- built here if missing):
# the lean GEMM kernels against the general kernel per shape / epilogue / pair, then a workgroup's life by phase for both.
R=${1:-r04}
[ -f mmnas_amd/lib/libmmnas_hip_stamp.so ] || make -C mmnas_amd/csrc variant NAME=stamp DEFS=-DMMNAS_DBG_STAMP=37 > /dev/null
{
echo "# Round 4: lean GEMM kernels (gemm_body<..., LEAN>; MMNAS_GEMM_LEAN) against the general kernel.  'general' = MMNAS_GEMM_LEAN=0, 'lean' = default (3)."
echo "# epi: b bias, R relu, d dropout 0.1, r residual, g gate (relu' / dropout replay), c column sums (lean level 3: element-wise epilogue), a accumulate onto C."
echo "# Whole-tile schedules: results asserted bit-equal; hybrid / stream-K schedules (K >= 512 here) and float-atomic weight gradients: max relative difference printed."
echo "# python tools/gemm_lean_ab.py   (one MI355X; us per launch = median of 5 blocks of 200 back-to-back launches; TF = algorithmic)"
python tools/gemm_lean_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175
echo
echo "# workgroup lifetime by phase, cycles (tools/gemm_stamps.py on a -DMMNAS_DBG_STAMP=37 build; 'epilogue' ends when the stores have landed)"
echo "## lean kernels (default)"
MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_stamp.so python tools/gemm_stamps.py 2>&1 | grep -v amdgpu.ids | grep "^NT\|lifetime" | tail -8
echo "## general kernel (MMNAS_GEMM_LEAN=0)"
MMNAS_GEMM_LEAN=0 MMNAS_LIB_PATH=$PWD/mmnas_amd/lib/libmmnas_hip_stamp.so python tools/gemm_stamps.py 2>&1 | grep -v amdgpu.ids | grep "^NT\|lifetime" | tail -8
} > profiles/${R}_gemm_lean_ab.txt
