#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run10
mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_ops_gpu.py tests/test_api_forms_gpu.py -x -q -m gpu -k "mha or att or rel" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
L=mmnas_amd/lib
echo "r2 build"; MMNAS_LIB_PATH=$PWD/$L/libmmnas_hip_r2.so python tools/mha_bench.py 2>/dev/null | grep "B="
echo "VPM 8 (default)"; python tools/mha_bench.py 2>/dev/null | grep "B="
for v in 5 12; do echo "VPM $v"; MMNAS_LIB_PATH=$PWD/$L/libmmnas_hip_vpm$v.so python tools/mha_bench.py 2>/dev/null | grep "B="; done
for wl in search_vqa train_vqa; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl', round(d['ms_per_step'],3), {k:(round(v['ms_per_step'],3), round(v['launches_per_step'],1)) for k,v in d['kernel_classes'].items()})
"
done
