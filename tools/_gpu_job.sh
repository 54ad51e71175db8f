set -o pipefail
mkdir -p gpurun_out
DAFS_HIP_DD_LISTS_WIDE=1 timeout -k 5 60 python tools/scratch/wide_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
DAFS_HIP_DD_LISTS_WIDE=1 timeout -k 10 600 python -m pytest tests/test_dd_gpu.py -x -q > gpurun_out/r3_dd_tests_wide.log 2>&1; rc=$?; tail -4 gpurun_out/r3_dd_tests_wide.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/scratch/c5_stages.py random 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_c5_random_stages_b.txt | head -9
exit 0
