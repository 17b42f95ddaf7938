mkdir -p gpurun_out/r6e
for v in 1 0 1 0 1 0; do
  MMNAS_MHA_BWD_B16=$v python -m pytest tests/test_dp_gpu.py -q -k "four_and_eight and supernet-8" > /tmp/t.log 2>&1; echo "B16=$v: $(tail -1 /tmp/t.log) $(grep -c ILLEGAL /tmp/t.log)"
done | tee gpurun_out/r6e/illegal_ab.txt
