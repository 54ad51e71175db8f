set -o pipefail
mkdir -p gpurun_out
./tools/scratch/malloc_time 2>&1 | tail -5
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r3_gpu_tests_full.log 2>&1; rc=$?; tail -25 gpurun_out/r3_gpu_tests_full.log
exit $rc
