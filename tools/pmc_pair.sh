#!/bin/bash
# PMC passes for the pair-HMM kernel on the headline workload (one rocprofv3 run per counter group,
# --kernel-trace only, as the pool requires).  Usage: [BENCH_FLAGS="--model contralign --config c5"] bash tools/pmc_pair.sh <outdir>
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$root/$out/p$i" -o p -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu --no-e2e $BENCH_FLAGS > "$root/$out/p$i.log" 2>&1 || echo "group $i failed"
done
cd "$root"
python3 tools/pmc_summary.py "$out"
