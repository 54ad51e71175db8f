python -m pytest tests -m gpu -x -q > gpurun_out/r2_gpu_tests7.log 2>&1 || { tail -30 gpurun_out/r2_gpu_tests7.log; exit 1; }
tail -3 gpurun_out/r2_gpu_tests7.log
VARIANTS="0 0 0" bash tools/bench_variants.sh gpurun_out/r2_variants7.log
BENCH_FLAGS="--n-seq 256 --length 200" VARIANTS="0 0 0" bash tools/bench_variants.sh gpurun_out/r2_variants7.log
BENCH_FLAGS="--n-seq 32 --length 80" VARIANTS="0 0 0" bash tools/bench_variants.sh gpurun_out/r2_variants7.log
BENCH_FLAGS="--model contralign" VARIANTS="0 0 0" bash tools/bench_variants.sh gpurun_out/r2_variants7.log
cat gpurun_out/r2_variants7.log
python bench.py > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err; cat gpurun_out/r2_bench7.json
