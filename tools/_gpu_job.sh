set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_cli_gpu.py -m gpu -x -q -k "devices" > gpurun_out/r3_cli_devices.log 2>&1; rc=$?; tail -30 gpurun_out/r3_cli_devices.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "workgroup_folding or wide_alignment or fault_shape" > gpurun_out/r3_wg_test.log 2>&1; rc=$?; tail -30 gpurun_out/r3_wg_test.log
[ $rc -eq 0 ] || exit $rc
DD_ROUNDS_ALL=1 timeout -k 10 300 python tools/dd_rounds5.py 512 400 family > gpurun_out/r3_rounds_c5_family.txt 2>&1; rc=$?; tail -5 gpurun_out/r3_rounds_c5_family.txt
[ $rc -eq 0 ] || exit $rc
DAFS_HIP_DD_WG=0 DD_ROUNDS_ALL=1 timeout -k 10 300 python tools/dd_rounds5.py 512 400 family > gpurun_out/r3_rounds_c5_family_nowg.txt 2>&1; rc=$?; tail -5 gpurun_out/r3_rounds_c5_family_nowg.txt
exit $rc
