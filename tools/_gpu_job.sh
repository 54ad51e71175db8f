set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_pct_gpu.py tests/test_dd_gpu.py -x -q > gpurun_out/r2_pct_tests.log 2>&1 || { tail -20 gpurun_out/r2_pct_tests.log; exit 1; }
tail -2 gpurun_out/r2_pct_tests.log
python tools/time_stages.py 128 150 > gpurun_out/r2_stages.log 2>&1; tail -6 gpurun_out/r2_stages.log
python tools/time_stages.py 256 200 > gpurun_out/r2_stages_c4.log 2>&1; tail -3 gpurun_out/r2_stages_c4.log
bash tools/pmc_stages.sh gpurun_out/pmc_stages_r2 > gpurun_out/pmc_stages_r2.log 2>&1; tail -40 gpurun_out/pmc_stages_r2.log
root=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r2_pc -o p -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu --no-e2e > $root/gpurun_out/prof_r2_pc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r2_ca -o p -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu --no-e2e --model contralign > $root/gpurun_out/prof_r2_ca.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r2_ca5 -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --model contralign --config c5 > $root/gpurun_out/prof_r2_ca5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r2_e2e -o p -- python3 $root/tools/time_stages.py 128 150 > $root/gpurun_out/prof_r2_e2e.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r2_forced -o p -- python3 $root/tools/time_forced.py > $root/gpurun_out/prof_r2_forced.log 2>&1
cd $root
tail -2 gpurun_out/prof_r2_pc.log gpurun_out/prof_r2_ca.log gpurun_out/prof_r2_ca5.log gpurun_out/prof_r2_forced.log | cut -c1-700
bash tools/pmc_pair.sh gpurun_out/pmc_r2c > gpurun_out/pmc_r2c.log 2>&1
BENCH_FLAGS="--model contralign" bash tools/pmc_pair.sh gpurun_out/pmc_r2ca > gpurun_out/pmc_r2ca.log 2>&1
BENCH_FLAGS="--model contralign --config c5" bash tools/pmc_pair.sh gpurun_out/pmc_r2ca5 > gpurun_out/pmc_r2ca5.log 2>&1
ls gpurun_out/prof_r2_pc gpurun_out/pmc_r2c | head
