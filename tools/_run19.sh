mkdir -p gpurun_out/r6g
MMNAS_COMMIT=$1 bash tools/refresh_profiles.sh r06 > gpurun_out/r6g/refresh.log 2>&1
tail -3 gpurun_out/r6g/refresh.log
