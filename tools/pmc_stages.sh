#!/bin/bash
# PMC pass over one whole run of the headline set (tools/time_stages.py): instruction mix and wait cycles of the
# folding and dual-decomposition kernels (--kernel-trace only, as the pool requires).  Usage: bash tools/pmc_stages.sh <outdir>
set -e
out=${1:-gpurun_out/pmc_stages}
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$root/$out/p$i" -o p -- python3 "$root/tools/time_stages.py" 128 150 > "$root/$out/p$i.log" 2>&1 || echo "group $i failed"
done
cd "$root"
python3 tools/pmc_stages_summary.py "$out"
