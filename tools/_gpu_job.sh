timeout -k 10 600 python -m pytest tests/test_dd_gpu.py tests/test_cli_gpu.py -x -q > gpurun_out/r2_dense_tests.log 2>&1; tail -12 gpurun_out/r2_dense_tests.log | cut -c1-300
grep -q "Memory access fault" gpurun_out/r2_dense_tests.log && exit 3
exit 0
