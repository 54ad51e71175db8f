"""c5-size round trace (CONTRAlign + CONTRAfold, N=512, L~400): like dd_rounds.py, one run, slow rounds only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import synth, pipeline, capi
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
res = pipeline.run(names, seqs, align_model=capi.ALIGN_CONTRALIGN)
print("seconds", {k: round(v, 3) for k, v in res.seconds.items()}, "rounds", res.levels)
for k, (dt, nodes) in enumerate(res.rounds):
    if dt > 0.02 or os.environ.get("DD_ROUNDS_ALL"):
        print("round %3d  %8.1f ms  open %3d  widths %s" % (k, dt * 1e3, len(nodes), sorted(max(a, b) for _, a, b in nodes)[-4:]))
print("sum of rounds %.2f s" % sum(dt for dt, _ in res.rounds))
