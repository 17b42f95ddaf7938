python -m pytest tests -q -m gpu -x 2>&1 | tail -4
for i in 1 2; do
for w in search_vqa arch_vqa search_vqa_unpad; do
for l in 0 3; do
echo "== $w lean=$l"; MMNAS_GEMM_LEAN=$l python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('avg_launch_us'))"
done; done; done
