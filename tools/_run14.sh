mkdir -p gpurun_out/r6e
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6e/pytest_all.log
for i in 1 2; do
for v in 1 0; do
MMNAS_MHA_BWD_B16=$v python bench.py --workload search_vqa --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']
print('search_vqa MMNAS_MHA_BWD_B16=$v %.4f ms/step (mha_bwd %.3f mha_fwd %.3f gemm %.3f)' % (d['ms_per_step'], k['mha_bwd'], k['mha_fwd'], k['gemm']))"
done; done | tee gpurun_out/r6e/b16_ab.txt
for v in 1 0; do
MMNAS_MHA_BWD_B16=$v python bench.py --workload train_vqa --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']
print('train_vqa MMNAS_MHA_BWD_B16=$v %.4f ms/step (mha_bwd %.3f mha_fwd %.3f gemm %.3f)' % (d['ms_per_step'], k['mha_bwd'], k['mha_fwd'], k['gemm']))"
done | tee -a gpurun_out/r6e/b16_ab.txt
