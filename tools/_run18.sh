mkdir -p gpurun_out/r6f
python -m pytest tests -q -m gpu > gpurun_out/r6f/pytest_all.log 2>&1; tail -5 gpurun_out/r6f/pytest_all.log
for i in 1 2; do
for v in "1 1" "0 0" "1 0"; do
set -- $v
MMNAS_MHA_BWD_B16=$1 MMNAS_MHA_FWD_B16=$2 python bench.py --workload search_vqa --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']
print('search_vqa BWD_B16=$1 FWD_B16=$2 %.4f ms/step (mha_bwd %.3f mha_fwd %.3f gemm %.3f rowops %.3f)' % (d['ms_per_step'], k['mha_bwd'], k['mha_fwd'], k['gemm'], k['rowops']))"
done; done | tee gpurun_out/r6f/b16_ab.txt
for wl in arch_vqa train_vqa; do for v in "1 1" "0 0"; do
set -- $v
MMNAS_MHA_BWD_B16=$1 MMNAS_MHA_FWD_B16=$2 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']
print('$wl BWD_B16=$1 FWD_B16=$2 %.4f ms/step (mha_bwd %.3f mha_fwd %.3f gemm %.3f)' % (d['ms_per_step'], k['mha_bwd'], k['mha_fwd'], k['gemm']))"
done; done | tee -a gpurun_out/r6f/b16_ab.txt
