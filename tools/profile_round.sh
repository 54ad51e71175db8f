#!/bin/bash
# The end-of-round evidence in one go: bench line, rocprofv3 kernel stats of the bench launch, of one whole run and of the
# forced-iteration leg, and the PMC passes (pair kernel with HBM traffic, stage kernels).  Usage: bash tools/profile_round.sh TAG
# Writes gpurun_out/<TAG>_*; copy what is to be judged into profiles/.
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_line.json 2> $out/${tag}_bench_line.err
tail -c 400 $out/${tag}_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_pc -o p -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-e2e > $out/prof_${tag}_pc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_e2e -o p -- python3 $root/tools/time_stages.py 128 150 > $out/prof_${tag}_e2e.log 2>&1
cd $root
cp $out/prof_${tag}_pc/p_kernel_stats.csv $out/${tag}_pairhmm3_kernel_stats.csv
cp $out/prof_${tag}_e2e/p_kernel_stats.csv $out/${tag}_e2e_kernel_stats.csv
bash tools/pmc_pair.sh gpurun_out/pmc_${tag}_pc > $out/pmc_${tag}_pc.log 2>&1
cp gpurun_out/pmc_${tag}_pc/summary.json $out/${tag}_pairhmm3_pmc.json
bash tools/pmc_stages.sh gpurun_out/pmc_${tag}_stages > $out/pmc_${tag}_stages.log 2>&1
cp gpurun_out/pmc_${tag}_stages/summary.json $out/${tag}_stage_kernels_pmc.json
head -8 $out/${tag}_e2e_kernel_stats.csv | cut -c1-120
