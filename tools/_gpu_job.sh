set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -x -q -s --durations=10 -k "rccl or whole_run_equals_oracle or c4_family_whole" > gpurun_out/r3_newtests.log 2>&1; rc=$?; tail -25 gpurun_out/r3_newtests.log
exit $rc
