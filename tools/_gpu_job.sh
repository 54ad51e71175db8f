set -x
tools/valu_rate > gpurun_out/r2_valu_rate2.log 2>&1; cat gpurun_out/r2_valu_rate2.log
python -m pytest tests/test_pairhmm_gpu.py -x -q > gpurun_out/r2_pair_tests2.log 2>&1 || { tail -30 gpurun_out/r2_pair_tests2.log; exit 1; }
tail -3 gpurun_out/r2_pair_tests2.log
VARIANTS="0 0 0;16 11 0;32 6 0;32 7 0;64 3 0" bash tools/bench_variants.sh gpurun_out/r2_variants2.log
BENCH_FLAGS="--n-seq 256 --length 200" VARIANTS="0 0 0;64 4 0" bash tools/bench_variants.sh gpurun_out/r2_variants2.log
cat gpurun_out/r2_variants2.log
