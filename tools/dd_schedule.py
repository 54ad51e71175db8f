"""How much of the progressive phase is lost to level-synchronous launches: sum over levels of the
slowest node versus the critical path of the guide tree (cost model: iterations+3 per node,
weighted by (L/150)^2).  Tuning aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dafs_amd import capi, synth, pipeline
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs])
score, left, right = res.tree
cost, level, finish = {}, {}, {}
for i in range(n):
    level[i] = 0; finish[i] = 0.0
for i in range(n, 2 * n - 1):
    it = res.dd_log[i][0]
    cost[i] = it + 3.0
    level[i] = max(level[left[i]], level[right[i]]) + 1
    finish[i] = max(finish[left[i]], finish[right[i]]) + cost[i]
lv = {}
for i, c in cost.items():
    lv[level[i]] = max(lv.get(level[i], 0.0), c)
print("levels", len(lv), "sum of level maxima", sum(lv.values()), "critical path", finish[2 * n - 2], "sum of all", sum(cost.values()))
print("per level max:", [int(lv[k]) for k in sorted(lv)])
