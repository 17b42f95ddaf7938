mkdir -p gpurun_out/r6e
for i in 1 2 3 4 5; do
  MMNAS_LSTM=0 python -m pytest tests/test_dp_gpu.py -q -k "four_and_eight and supernet-8" > /tmp/t.log 2>&1; echo "LSTM=0: $(tail -1 /tmp/t.log) $(grep -c ILLEGAL /tmp/t.log)"
done | tee gpurun_out/r6e/illegal_lstm0.txt
for i in 1 2 3; do
  MMNAS_SMALL_OPS=0 python -m pytest tests/test_dp_gpu.py -q -k "four_and_eight and supernet-8" > /tmp/t.log 2>&1; echo "SMALL_OPS=0: $(tail -1 /tmp/t.log) $(grep -c ILLEGAL /tmp/t.log)"
done | tee -a gpurun_out/r6e/illegal_lstm0.txt
