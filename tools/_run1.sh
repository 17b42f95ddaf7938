set -u
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or mha or attention" > gpurun_out/r6a/pytest_kernels.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/pytest_kernels.log
tail -3 gpurun_out/r6a/pytest_kernels.log
python tools/ln_bench.py > gpurun_out/r6a/ln_bench.txt 2>&1; cat gpurun_out/r6a/ln_bench.txt
python tools/mha_bench.py > gpurun_out/r6a/mha_bench.txt 2>&1; cat gpurun_out/r6a/mha_bench.txt
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ln -o t -- python3 $R/tools/ln_bench.py > /tmp/p_ln.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mha -o t -- python3 $R/tools/mha_bench.py > /tmp/p_mha.log 2>&1)
for f in $(find /tmp/p_ln /tmp/p_mha -name "*kernel_stats.csv"); do echo == $f; head -12 $f | cut -c1-200; cp $f gpurun_out/r6a/$(echo $f | tr '/' '_'); done
