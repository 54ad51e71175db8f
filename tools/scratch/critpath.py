"""Critical path of the progressive phase in time (iterations x a per-iteration cost that grows with the node) against
what the round-based driver takes.  Tuning aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth, pipeline
n, L = int(sys.argv[1]), int(sys.argv[2])
recs = synth.random_set(n, L, seed=12345)
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs])
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs])
score, left, right = res.tree
fin = {i: 0.0 for i in range(n)}
per_it = float(os.environ.get("PER_IT_US", "83"))
setup = float(os.environ.get("SETUP_US", "400"))
for i in range(n, 2 * n - 1):
    it = res.dd_log[i][0]
    d = res.dd_log[i][1] if len(res.dd_log[i]) > 1 else None
    fin[i] = max(fin[left[i]], fin[right[i]]) + setup + it * per_it
print("progressive %.1f ms; critical path %.1f ms (per iteration %.0f us, per node %.0f us); rounds %d" % (1e3 * res.seconds["progressive"], fin[2 * n - 2] / 1e3, per_it, setup, len(res.rounds) if getattr(res, "rounds", None) else -1))
# the chain
i = 2 * n - 2
chain = []
while i >= n:
    chain.append((i, res.dd_log[i][0]))
    i = left[i] if fin[left[i]] >= fin[right[i]] else right[i]
print("chain (node, iterations):", chain)
# slack of every node with many iterations: how much later its sibling's side finishes
par = {}
for i in range(n, 2 * n - 1):
    par[left[i]] = i; par[right[i]] = i
for i in range(n, 2 * n - 1):
    if res.dd_log[i][0] >= 100:
        # walk up: slack = min over ancestors of (finish of the other child - finish of this side)
        j, slack = i, None
        while j in par:
            pj = par[j]
            other = right[pj] if left[pj] == j else left[pj]
            d = fin[other] - fin[j]
            slack = d if slack is None else max(slack, d)
            if d > 0: break
            j = pj
        print("node %d iters %d: finishes at %.1f ms; first ancestor where the other side is later: +%.1f ms" % (i, res.dd_log[i][0], fin[i] / 1e3, (slack or 0) / 1e3))
