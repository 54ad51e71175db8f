timeout -k 10 600 python -m pytest tests/test_dd_gpu.py -x -q > gpurun_out/r2_dd_tests10.log 2>&1; tail -15 gpurun_out/r2_dd_tests10.log | cut -c1-300
grep -q "Memory access fault" gpurun_out/r2_dd_tests10.log && exit 3
python tools/time_stages.py 128 150 2>&1 | tail -3
exit 0
