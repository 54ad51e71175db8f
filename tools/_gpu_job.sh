set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dd_gpu.py -x -q --durations=5 > gpurun_out/r3_dd_tests.log 2>&1; rc=$?; tail -12 gpurun_out/r3_dd_tests.log
[ $rc -ne 0 ] && exit $rc
python tools/time_stages.py 128 150 2>&1 | grep -v amdgpu.ids | tail -3
DAFS_HIP_DD_SPAN_MW=0 python tools/time_stages.py 128 150 2>&1 | grep -v amdgpu.ids | tail -3
exit 0
