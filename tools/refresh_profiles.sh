#!/bin/bash
# Regenerates everything under profiles/ for one round on a GPU box (run from the repo root):
#   MMNAS_COMMIT=$(git rev-parse --short=12 HEAD) bash tools/refresh_profiles.sh r05      (MMNAS_COMMIT: stamped into the PMC files)
# rocprofv3 --kernel-trace --stats summaries of the bench command per workload, one-step kernel timelines, the two
# --pmc passes behind roofline.traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass), the MFMA / LDS counter passes, and
# the bench lines themselves (the default line with its CPU baselines last: it reads the traffic files written before).
# Raw traces stay in /tmp; only summaries are written to profiles/.  Counters are collected with --kernel-trace only.
set -u
R=${1:-r06}
ROOT=$PWD
export TMPDIR=/tmp
W=/tmp/mmnas_prof
rm -rf $W; mkdir -p $W profiles
for wl in search_vqa arch_vqa train_vqa search_vqa_unpad train_vqa_unpad; do
  cmd="bench.py --workload $wl --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-fixed-cost"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace_$wl -o t -- python3 $ROOT/$cmd > $W/trace_$wl.log 2>&1)
  SUMMARIZE_PASSES=3,10,3,10 python3 tools/summarize_prof.py $W/trace_$wl profiles/${R}_$wl auto \
    "rocprofv3 --kernel-trace --stats -- python3 $cmd  (3 warm-up + 10 timed + 3 empty-queue + 10 roofline-pass steps)"
  marker=onehot_rows; [ $wl = train_vqa ] && marker=row_is_zero; [ $wl = train_vqa_unpad ] && marker=row_is_zero
  python3 tools/step_timeline.py $W/trace_$wl $marker > profiles/${R}_timeline_$wl.txt
  [ -n "${SKIP_PMC:-}" ] && continue
  case $wl in *_unpad) continue;; esac   # (the ragged records: kernel statistics and timeline only)
  small="bench.py --workload $wl --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-prof --no-fixed-cost"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $W/pmc_${wl}_$c -o t -- python3 $ROOT/$small > $W/pmc_${wl}_$c.log 2>&1)
  done
  python3 tools/pmc_traffic.py $W/pmc_${wl}_FETCH_SIZE $W/pmc_${wl}_WRITE_SIZE profiles/${R}_traffic_$wl.json
  python3 tools/stamp_profiles.py profiles/${R}_traffic_$wl.json > /dev/null   # (before the bench lines below read it)
  [ $wl = arch_vqa ] && continue
  (cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $W/pmc_${wl}_mfma -o t -- python3 $ROOT/$small > $W/pmc_${wl}_mfma.log 2>&1)
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $W/pmc_${wl}_lds -o t -- python3 $ROOT/$small > $W/pmc_${wl}_lds.log 2>&1)
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $W/pmc_${wl}_gui -o t -- python3 $ROOT/$small > $W/pmc_${wl}_gui.log 2>&1)
  python3 tools/pmc_counters.py profiles/${R}_pmc_$wl.json $W/pmc_${wl}_mfma $W/pmc_${wl}_lds $W/pmc_${wl}_gui
  python3 tools/stamp_profiles.py profiles/${R}_pmc_$wl.json > /dev/null
done
# the data-parallel exchange in a one-rank RCCL group, and the streamed-input step with the copy engine's rows
for wl in search_vqa_dp1 train_vqa_dp1; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $W/trace_$wl -o t -- python3 $ROOT/bench.py --workload $wl --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-prof > $W/trace_$wl.log 2>&1)
  marker=onehot_rows; [ $wl = train_vqa_dp1 ] && marker=row_is_zero
  python3 tools/step_timeline.py $W/trace_$wl $marker --list > profiles/${R}_timeline_$wl.txt
done
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $W/trace_stream -o t -- python3 $ROOT/bench.py --workload search_vqa_stream --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline --no-prof > $W/trace_stream.log 2>&1)
python3 tools/copy_overlap.py $W/trace_stream > profiles/${R}_timeline_search_vqa_stream.txt 2>&1
if [ -z "${SKIP_PMC:-}" ]; then
  small="bench.py --workload bilevel_vqa --steps 6 --warmup 6 --repeats 1 --no-cpu-baseline --no-prof --no-fixed-cost"   # (one round = 5 weight + 1 arch steps)
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $W/pmc_bilevel_vqa_$c -o t -- python3 $ROOT/$small > $W/pmc_bilevel_vqa_$c.log 2>&1)
  done
  python3 tools/pmc_traffic.py $W/pmc_bilevel_vqa_FETCH_SIZE $W/pmc_bilevel_vqa_WRITE_SIZE profiles/${R}_traffic_bilevel_vqa.json
  python3 tools/stamp_profiles.py profiles/${R}_traffic_bilevel_vqa.json > /dev/null
  (cd /tmp && GEMM_PMC_SWEEP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $W/sweep -o t -- python3 $ROOT/tools/gemm_pmc.py > $W/sweep.log 2>&1)
  python3 tools/traffic_sweep.py $W/sweep > profiles/${R}_gemm_traffic_sweep.txt
fi
python3 tools/mha_bench.py > profiles/${R}_mha_microbench.txt 2>/dev/null
python3 tools/rel_bench.py > profiles/${R}_rel_microbench.txt 2>/dev/null
# the unchanged search loop's host cost, with the library's one-node zero terms (default) and with the literal lines
{ echo "# MMNAS_ZERO_TERMS=1 (default: SumParameter / LazySum, mmnas_amd/zeroterm.py)"; python3 tools/dropin_host_split.py 2>/dev/null
  echo; echo "# MMNAS_ZERO_TERMS=0 (plain parameters: the three 0 * sum(p.sum()) lines run as written)"; MMNAS_ZERO_TERMS=0 python3 tools/dropin_host_split.py 2>/dev/null; } > profiles/${R}_host_dropin.txt
python3 tools/dropin_host_split.py --statements > profiles/${R}_host_dropin_statements.txt 2>/dev/null
python3 tools/dropin_host_split.py --floor > profiles/${R}_host_dropin_floor.txt 2>/dev/null
python3 tools/ln_bench.py > profiles/${R}_ln_microbench.txt 2>/dev/null
python3 tools/gemm_ln_bench.py > profiles/${R}_gemm_ln_microbench.txt 2>/dev/null
[ -f mmnas_amd/lib/libmmnas_hip_fwd1.so ] && bash tools/mha_fwd_phases.sh > profiles/${R}_mha_fwd_phases.txt 2>&1
[ -z "${SKIP_AB:-}" ] && bash tools/r06_ab.sh > profiles/${R}_ab.txt 2>&1
[ -n "${OLD_AB:-}" ] && bash tools/hoist_ab.sh > profiles/${R}_hoist_ab.txt 2>&1       # (round 5's A/Bs: OLD_AB=1)
[ -n "${OLD_AB:-}" ] && bash tools/small_bwd_ab.sh > profiles/${R}_small_bwd_ab.txt 2>&1
(cd /tmp && GEMM_PMC_LAYOUTS=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $W/pmc_gemm_lds -o t -- python3 $ROOT/tools/gemm_pmc.py > $W/pmc_gemm_lds.log 2>&1)
(cd /tmp && GEMM_PMC_LAYOUTS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $W/pmc_gemm_mfma -o t -- python3 $ROOT/tools/gemm_pmc.py > $W/pmc_gemm_mfma.log 2>&1)
python3 tools/pmc_counters.py profiles/${R}_pmc_gemm_layouts.json $W/pmc_gemm_lds $W/pmc_gemm_mfma
python3 tools/stamp_profiles.py profiles/${R}_pmc_gemm_layouts.json > /dev/null
# the bench lines: <round>_bench.json = the FULL record (bench.py --full-out), <round>_bench_line.json = the compact stdout line
python3 bench.py --full-out profiles/${R}_bench.json > $W/bench_all.log 2> $W/bench_all.err
grep '^{' $W/bench_all.log | tail -1 > profiles/${R}_bench_line.json
for wl in train_vgd train_itm; do
  python3 bench.py --workload $wl --no-cpu-baseline --full-out profiles/${R}_bench_$wl.json > /dev/null 2>&1
done
# the same steps with the matrix products on the fp32 MFMA (MMNAS_GEMM_SPLIT=0) and as 3 bf16 products (experiment)
for sp in 0 3; do for wl in search_vqa train_vqa; do
  python3 bench.py --workload $wl --no-cpu-baseline --gemm-split $sp --full-out profiles/${R}_bench_${wl}_gemm_split$sp.json > /dev/null 2>&1
done; done
# BASELINE configs[4] ("fp16 MFMA"): the single-pass bf16 flavour (reduced precision, tolerance 3e-2: tests/test_harness_gpu.py)
python3 bench.py --workload train_itm --no-cpu-baseline --gemm-split 1 --full-out profiles/${R}_bench_train_itm_gemm_split1.json > /dev/null 2>&1
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/   # (gpurun merges only gpurun_out/ back)
ls -la profiles/ | grep $R
for f in $W/*.log; do echo "== $f"; tail -n 2 $f | cut -c1-300; done
