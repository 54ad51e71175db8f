set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py -x -q -s --durations=12 -k "fault_shape or failed_open or arena" > gpurun_out/r3_fault_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r3_fault_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_dd_gpu.py -x -q --durations=8 > gpurun_out/r3_dd_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r3_dd_tests.log
exit $rc
