set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py -x -q --durations=8 > gpurun_out/r3_dist_tests.log 2>&1; rc=$?; tail -30 gpurun_out/r3_dist_tests.log
exit $rc
