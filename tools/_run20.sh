mkdir -p gpurun_out/r6h
python -m pytest tests/test_kernels_gpu.py tests/test_chain_gpu.py tests/test_nets_gpu.py tests/test_nets_full_gpu.py -x -q 2>&1 | tail -4
python tools/mha_bench.py 2>/dev/null | head -4
for i in 1 2; do for v in 1 0; do for wl in arch_vqa train_vqa; do
MMNAS_MHA_FWD_B16=$v python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']
print('$wl FWD_B16=$v %.4f ms/step (mha_bwd %.3f mha_fwd %.3f gemm %.3f)' % (d['ms_per_step'], k['mha_bwd'], k['mha_fwd'], k['gemm']))"
done; done; done | tee gpurun_out/r6h/fwd16_two_ab.txt
