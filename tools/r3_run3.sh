#!/bin/bash
# round-3 GPU call 3: whole GPU suite on the new build, full bench line
set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r3_run3
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/test_gpu.log 2>&1; echo "gpu suite rc=$?" >> $O/test_gpu.log
timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err; echo "bench rc=$?" >> $O/bench_all.err
tail -5 $O/test_gpu.log; tail -3 $O/bench_all.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3_run3/bench_all.json') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], d.get('value_min'), d.get('value_max'), d['host_issue_ms_per_step'], d['roofline']['frac'])
for k,v in d['sub'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('value_min'), v.get('value_max'), v.get('host_issue_ms_per_step'), v.get('library_launches_per_step'), v.get('ms_per_step_vs_plain'), v.get('error'))
PY
