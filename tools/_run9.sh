mkdir -p gpurun_out/r6a
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r6a/pytest_all9.log
python bench.py --workload search_vqa --no-cpu-baseline --full-out gpurun_out/r6a/bench9_on.json > gpurun_out/r6a/bench9_on.line 2>/dev/null
MMNAS_GEMM_LN=0 python bench.py --workload search_vqa --no-cpu-baseline --full-out gpurun_out/r6a/bench9_off.json > gpurun_out/r6a/bench9_off.line 2>/dev/null
python bench.py --workload search_vqa --no-cpu-baseline --full-out gpurun_out/r6a/bench9_on2.json > gpurun_out/r6a/bench9_on2.line 2>/dev/null
MMNAS_GEMM_LN=0 python bench.py --workload search_vqa --no-cpu-baseline --full-out gpurun_out/r6a/bench9_off2.json > gpurun_out/r6a/bench9_off2.line 2>/dev/null
for f in on off on2 off2; do python -c "
import json,sys
d=json.load(open('gpurun_out/r6a/bench9_$f.json'))
print('$f', d['ms_per_step'], d.get('blocks_ms_per_step'), d['kernel_ms_per_step'] if 'kernel_ms_per_step' in d else '')
"; done
