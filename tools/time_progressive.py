"""Where the wall time of the progressive phase goes on the host side (tuning aid): time inside the C calls
nodes_open / nodes_advance / nodes_result versus the Python bookkeeping around them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth, pipeline
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
ctx = capi.Context(0)
pipeline.run(names, seqs, ctx=ctx)
acc = {}
def wrap(name):
    f = getattr(capi.Context, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        acc[name + "#"] = acc.get(name + "#", 0) + 1
        return r
    setattr(capi.Context, name, g)
for nm in ("nodes_round", "nodes_open", "nodes_advance", "nodes_result", "nodes_close"):
    wrap(nm)
res = pipeline.run(names, seqs, ctx=ctx)
print("progressive %.1f ms" % (1e3 * res.seconds["progressive"]), {k: (round(1e3 * v, 1) if not k.endswith("#") else v) for k, v in acc.items()})
