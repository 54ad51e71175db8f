set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_pct_gpu.py -x -q > gpurun_out/r3_pct_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r3_pct_tests.log
[ $rc -ne 0 ] && exit $rc
python tools/scratch/pct_time.py 128 150 2>&1 | grep -v amdgpu.ids | tail -1
python tools/scratch/pct_time.py 256 200 2>&1 | grep -v amdgpu.ids | tail -1
python tools/scratch/pct_time.py 128 150 family 2>&1 | grep -v amdgpu.ids | tail -1
DAFS_HIP_PCT_GRID2D=1 python tools/scratch/pct_time.py 128 150 2>&1 | grep -v amdgpu.ids | tail -1
exit $rc
