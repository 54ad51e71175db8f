set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "workgroup_folding or fault_shape" > gpurun_out/r3_wg3_test.log 2>&1; rc=$?; tail -5 gpurun_out/r3_wg3_test.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 bash tools/profile_round.sh r03_b; rc=$?
exit $rc
