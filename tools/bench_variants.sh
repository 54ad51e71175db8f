#!/bin/bash
# A/B of pair-HMM kernel variants on the headline workload (tuning aid)
run() { env "$@" python bench.py --steps 10 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', '|', d['config']['kernel'], round(d['roofline']['kernel_ms'],3), 'ms', round(d['value']/1e6,3), 'Mpairs/s')"; }
run A=0
run DAFS_HIP_FORCE_GROUP=32 DAFS_HIP_WAVES_PER_SIMD=4
run DAFS_HIP_FORCE_GROUP=32 DAFS_HIP_WAVES_PER_SIMD=3
run DAFS_HIP_FORCE_GROUP=64 DAFS_HIP_WAVES_PER_SIMD=8
run DAFS_HIP_FORCE_GROUP=64 DAFS_HIP_WAVES_PER_SIMD=4
