mkdir -p gpurun_out/r6k
for i in 1 2 3 4 5 6; do python -m pytest tests/test_bench_gpu.py -x -q -k default_line > /tmp/t$i.log 2>&1; tail -1 /tmp/t$i.log; if grep -q "1 failed" /tmp/t$i.log; then cp /tmp/t$i.log gpurun_out/r6k/fail_$i.log; grep -n "^E " /tmp/t$i.log | head -20; fi; done
