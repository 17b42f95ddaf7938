#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run11
mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_chain_gpu.py -x -q -m gpu -k "rel or nets or chain" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
echo "VALU kernel"; python tools/rel_bench.py 2>/dev/null | grep "B="
echo "MFMA kernel"; MMNAS_REL_BWD_VALU=0 python tools/rel_bench.py 2>/dev/null | grep "B="
for wl in search_vqa; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --repeats 5 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl', round(d['ms_per_step'],3), {k:(round(v['ms_per_step'],3), round(v['launches_per_step'],1)) for k,v in d['kernel_classes'].items()})
"
done
