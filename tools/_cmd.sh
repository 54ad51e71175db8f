set -e
python -m pytest tests/test_dd_gpu.py -x -q > gpurun_out/r2b_dd_tests2.log 2>&1 || { tail -30 gpurun_out/r2b_dd_tests2.log; exit 1; }
tail -2 gpurun_out/r2b_dd_tests2.log
python tools/dd_rounds.py 128 150 > gpurun_out/r2b_rounds_c3_span2.log 2>&1
head -3 gpurun_out/r2b_rounds_c3_span2.log; grep "advance total" gpurun_out/r2b_rounds_c3_span2.log
DAFS_HIP_DD_STAMPS=1 python tools/time_progressive.py 128 150 > gpurun_out/r2b_stamps_c3_span2.log 2>&1
grep "iters=6" gpurun_out/r2b_stamps_c3_span2.log | head
python tools/e2e_check.py 128 150 random cmp > gpurun_out/r2b_e2e_check.log 2>&1; tail -1 gpurun_out/r2b_e2e_check.log
