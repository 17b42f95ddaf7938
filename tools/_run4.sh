set -u
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_kernels_gpu.py tests/test_runtime_config.py -x -q > gpurun_out/r6a/pytest4.log 2>&1; tail -3 gpurun_out/r6a/pytest4.log
python tools/mha_bench.py 2>/dev/null | tee gpurun_out/r6a/mha_bench4.txt
prof() {
  (cd /tmp && rm -rf /tmp/pp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 $R/tools/ln_bench.py $2 > /tmp/pp.log 2>&1)
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  python3 - "$1" "$2" "$f" <<'PY'
import csv,sys
lab,shape,f=sys.argv[1:4]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'ln_bwd' in n or 'ln_fwd' in n:
        print('%-10s %-14s %-40s calls %4s avg %7.2f us min %6.2f us' % (lab, shape, n.split('(')[0].replace('void mmnas::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
}
for i in 1 2; do for shp in "6400 256 0.1" "6400 512 0.1"; do
  MMNAS_LIB_PATH=$R/mmnas_amd/lib/libmmnas_hip_oldln.so prof old "$shp"
  prof new "$shp"
done; done 2>&1 | grep ln_bwd_kernel | tee gpurun_out/r6a/ln_ab4.txt
