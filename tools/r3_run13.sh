#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run13
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/test_gpu.log 2>&1; echo "gpu suite rc=$?" >> $O/test_gpu.log
tail -6 $O/test_gpu.log
show='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],3), round(1000/d["value_max"],3), round(1000/d["value_min"],3), round(d["host_issue_ms_per_step_empty_queue"],2))
        kc=d.get("kernel_classes")
        if kc: print("   ", {k:(round(v["ms_per_step"],3), round(v["launches_per_step"],1)) for k,v in kc.items()})
'
for wl in arch_vqa bilevel_vqa search_vqa train_vqa; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "$show" $wl
done
timeout 600 python tools/gemm_ab.py mmnas_amd/lib/libmmnas_hip_r2.so mmnas_amd/lib/libmmnas_hip.so 2>/dev/null | awk '{print $1,$2,$3,$4,$5, $(NF-5), $(NF-4), $(NF)}' | head -30
