"""Per-kernel sums of the PMC values written by tools/pmc_stages.sh (all dispatches of a kernel in the run added up)."""
import csv, glob, json, os, sys
out = sys.argv[1]
acc = {}
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            for key in ("k_dd_solve", "k_contrafold_posterior", "k_contrafold", "k_pct_rows", "k_node_lists", "k_node_avg"):
                if key in name:
                    d = acc.setdefault(key, {})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    d.setdefault("_dispatches", set()).add(row["Dispatch_Id"])
                    break
res = {k: {c: (len(v) if c == "_dispatches" else v) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(res, indent=1, sort_keys=True))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
