#!/bin/bash
# A/B of pair-HMM kernel variants (tuning aid).  One line per variant, appended to OUTFILE.
#   VARIANTS="G W OCC;G W OCC;..." BENCH_FLAGS="..." bash tools/bench_variants.sh OUTFILE      (0 = planner's choice)
out=$1
VARIANTS=${VARIANTS:-"0 0 0;16 0 0;32 0 0;64 0 0"}
IFS=';' read -ra vs <<< "$VARIANTS"
for v in "${vs[@]}"; do
  read g w o <<< "$v"
  env DAFS_HIP_FORCE_GROUP=$g DAFS_HIP_FORCE_WIDTH=$w DAFS_HIP_WAVES_PER_SIMD=$o \
    python bench.py --steps 10 --warmup 2 --no-cpu --no-e2e $BENCH_FLAGS 2>>$out.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('G=$g W=$w OCC=$o $BENCH_FLAGS', '|', d['config']['kernel'], round(d['roofline']['kernel_ms'],3), 'ms', round(d['value']/1e6,3), 'Mpairs/s', 'frac', round(d['roofline']['frac'],3))" >> $out
done
