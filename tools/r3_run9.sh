#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run9
mkdir -p $O
bash tools/mha_phases.sh > $O/mha_phases.txt 2>&1
cat $O/mha_phases.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/test_gpu.log 2>&1; echo "gpu suite rc=$?" >> $O/test_gpu.log
tail -4 $O/test_gpu.log
timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err; echo "bench rc=$?" >> $O/bench_all.err
tail -2 $O/bench_all.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3_run9/bench_all.json') if l.startswith('{')][-1])
print('headline', round(d['value'],2), round(d['ms_per_step'],3), round(d.get('value_min'),1), round(d.get('value_max'),1), 'host', round(d['host_issue_ms_per_step'],2), round(d['host_issue_ms_per_step_empty_queue'],2), 'frac', round(d['roofline']['frac'],3))
for k,v in d['sub'].items():
    print(k, v.get('value') and round(v['value'],2), v.get('ms_per_step') and round(v['ms_per_step'],3), 'host', v.get('host_issue_ms_per_step') and round(v['host_issue_ms_per_step'],2), v.get('host_issue_ms_per_step_empty_queue') and round(v['host_issue_ms_per_step_empty_queue'],2), v.get('library_launches_per_step'), v.get('ms_per_step_vs_plain'), v.get('error'))
PY
