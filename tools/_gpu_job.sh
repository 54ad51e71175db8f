timeout -k 10 600 python -m pytest tests/test_dist_gpu.py tests/test_dd_gpu.py tests/test_pct_gpu.py -x -q > gpurun_out/r2_dist_tests.log 2>&1; tail -30 gpurun_out/r2_dist_tests.log | cut -c1-400
grep -q "Memory access fault" gpurun_out/r2_dist_tests.log && exit 3
exit 0
