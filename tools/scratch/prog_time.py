import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth, pipeline
n, L = int(sys.argv[1]), int(sys.argv[2])
recs = synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
ctx = capi.Context(0)
best = None
for rep in range(int(os.environ.get("REPS", "5"))):
    res = pipeline.run(names, seqs, ctx=ctx)
    p = res.seconds["progressive"]; t = res.seconds["total"]
    if rep and (best is None or t < best[1]): best = (p, t, len(res.rounds))
print("wake %s: progressive %.1f ms total %.1f ms rounds %d" % (os.environ.get("DAFS_HIP_DD_WAKE", "default"), 1e3 * best[0], 1e3 * best[1], best[2]))
