set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dd_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r3_panels_test.log 2>&1; rc=$?; tail -30 gpurun_out/r3_panels_test.log
[ $rc -eq 0 ] || exit $rc
DAFS_HIP_DD_STAMPS=1 timeout -k 10 300 python tools/dd_rounds5.py 512 400 family > gpurun_out/r3_stamps_c5_family.txt 2> gpurun_out/r3_stamps_c5_family.err; rc=$?; tail -3 gpurun_out/r3_stamps_c5_family.txt
grep "dd node" gpurun_out/r3_stamps_c5_family.err | awk '{ if (substr($3,4)+0 > 600 || substr($4,4)+0 > 600) print }' | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/dd_rounds5.py 512 400 > gpurun_out/r3_rounds_c5_random_c.txt 2>&1; rc=$?; tail -25 gpurun_out/r3_rounds_c5_random_c.txt
exit $rc
