"""Where the progressive phase spends its time: per round of the resident-node schedule, the launch time and the
widths of the nodes that were open (tuning aid).   python tools/dd_rounds.py N L [family]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dafs_amd import synth, pipeline
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
pipeline.run(names, seqs)
res = pipeline.run(names, seqs)
print("seconds", {k: round(v, 4) for k, v in res.seconds.items()}, "rounds", res.levels)
tot = 0.0
edges = [0, 200, 230, 330, 415, 512, 768, 1 << 20]
by = {e: 0.0 for e in edges[1:]}
for k, (dt, nodes) in enumerate(res.rounds):
    w = max(max(a, b) for _, a, b in nodes)
    tot += dt
    for e in edges[1:]:
        if w <= e:
            by[e] += dt
            break
    print("round %2d  %6.2f ms  open %3d  widest %4d  widths %s" % (k, dt * 1e3, len(nodes), w, sorted(max(a, b) for _, a, b in nodes)[-6:]))
print("advance total %.1f ms; by widest open node:" % (tot * 1e3), {("<=%d" % e): round(v * 1e3, 1) for e, v in by.items()})
for i in sorted(res.dd_dims):
    print("node", i, "dims", res.dd_dims[i], "iters", res.dd_log[i][0], "ncbp", res.dd_log[i][2])
