set -u
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
R=$PWD
python tools/mha_bench.py > gpurun_out/r6a/mha_bench2.txt 2>&1; cat gpurun_out/r6a/mha_bench2.txt
for lib in new old; do
  [ $lib = old ] && export MMNAS_LIB_PATH=$R/mmnas_amd/lib/libmmnas_hip_oldln.so
  echo "== $lib"; python tools/ln_bench.py 2>/dev/null
  for shp in "6400 256 0.1" "896 256 0.1" "6400 512 0.1"; do
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ln_$lib -o t -- python3 $R/tools/ln_bench.py $shp > /tmp/p_ln.log 2>&1)
    echo "-- $lib $shp"; grep "ln_" /tmp/p_ln_$lib/*/t_kernel_stats.csv | cut -d, -f1-4 | sed 's/(float const.*)"/"/' | cut -c1-160
    rm -rf /tmp/p_ln_$lib
  done
done 2>&1 | tee gpurun_out/r6a/ln_ab.txt
