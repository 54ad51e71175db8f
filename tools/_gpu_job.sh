timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r2_gpu_tests8.log 2>&1; tail -22 gpurun_out/r2_gpu_tests8.log
grep -q "Memory access fault" gpurun_out/r2_gpu_tests8.log && exit 3
exit 0
