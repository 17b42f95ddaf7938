mkdir -p gpurun_out/r6b
python tools/gemm_ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6b/gemm_ln_bench.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r6b/pytest_all.log
python bench.py --full-out gpurun_out/r6b/bench_full.json > gpurun_out/r6b/bench.line 2> gpurun_out/r6b/bench.err; tail -c 3500 gpurun_out/r6b/bench.line
