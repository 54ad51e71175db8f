"""Summarise the per-node phase timings that DAFS_HIP_DD_STAMPS=1 prints to stderr (tuning aid).
usage: python tools/dd_stamps.py stderr.log"""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"dd node L1=(\d+) L2=(\d+) n=(\d+)\+(\d+) ncbp=(\d+) iters=(\d+) slow-xy=(\d+)\+(\d+) \| us: x-dp (\d+) x-traceback (\d+) wait (\d+) cbp (\d+) update (\d+) tail (\d+)", l)
    if m:
        rows.append(list(map(int, m.groups())))
it = max(sum(r[5] for r in rows), 1)
print("nodes", len(rows), "iters", it, "per-iter us: xdp %.1f tb %.1f wait %.1f cbp %.1f upd %.1f tail %.1f" % tuple(sum(r[k] for r in rows) / it for k in range(8, 14)))
