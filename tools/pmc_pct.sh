#!/bin/bash
# PMC passes over the consistency stage of the headline set (tools/time_pct.py): instruction mix, waits, LDS conflicts and
# the L1 / L2 request counters of k_pct_rows (--kernel-trace only, one pass per counter group).  Usage: bash tools/pmc_pct.sh <outdir> [N L]
out=${1:-gpurun_out/pmc_pct}
n=${2:-128}; l=${3:-150}
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$root/$out/p$i" -o p -- python3 "$root/tools/time_pct.py" $n $l > "$root/$out/p$i.log" 2>&1 || echo "group $i failed: $grp"
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
acc = {}
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            for key in ("k_pct_rows", "k_pct_bp_rows", "k_pct_emit"):
                if key in name and not ("k_pct_bp" in name and key == "k_pct_rows"):
                    d = acc.setdefault(key, {})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    d.setdefault("_dispatches", set()).add(row["Dispatch_Id"])
                    break
res = {k: {c: (len(v) if c == "_dispatches" else v) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(res.get("k_pct_rows", {}), indent=1, sort_keys=True))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
PY
