python -m pytest tests/test_chain_gpu.py -q -x -k overlap 2>&1 | tail -2
python tools/gemm_knob.py MMNAS_GEMM_PF 1 2 2>&1 | grep "6400" | cut -c1-130 | head -30
for i in 1 2; do for pf in 1 2; do for wl in search_vqa train_vqa; do
MMNAS_GEMM_PF=$pf python bench.py --workload $wl --steps 20 --no-cpu-baseline --no-prof > gpurun_out/b.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/b.log'):
    if l.startswith('{'):
        d=json.loads(l); print('pf=$pf $wl', round(d['value'],1), round(d['ms_per_step'],3))
PY
done; done; done
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -2
