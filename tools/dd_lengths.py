"""Alignment widths met by the progressive phase and the iterations spent at each width (tuning aid: which
LDS placement of the folding DPs the nodes of a workload fall into)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dafs_amd import synth, pipeline
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
model = 1 if "contralign" in sys.argv[3:] else 0
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs], align_model=model)
print("seconds", {k: round(v, 3) for k, v in res.seconds.items()}, "rounds", res.levels)
w = np.array([max(res.dd_dims[i]) for i in sorted(res.dd_dims)])
it = np.array([res.dd_log[i][0] for i in sorted(res.dd_dims)])
edges = [0, 230, 250, 330, 415, 448, 480, 512, 640, 1024, 4096, 16384, 1 << 20]
for a, b in zip(edges[:-1], edges[1:]):
    m = (w > a) & (w <= b)
    if m.any():
        print("width (%d, %d]: %d nodes, iterations sum %d max %d" % (a, b, m.sum(), it[m].sum(), it[m].max()))
print("widest", w.max(), "root", res.dd_dims[2 * n - 2], "node memory (reserved, in use, peak) MB", [m >> 20 for m in res.dd_memory])
