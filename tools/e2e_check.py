import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib
from dafs_amd import synth, pipeline, capi
from test_pct_gpu import random_bp
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
bp = random_bp(seqs, 1, density=0.02)
ctx = capi.Context(0)
pipeline.run(names[:8], seqs[:8], ctx=ctx, bp=bp[:8])  # warm-up
t0 = time.perf_counter()
got = pipeline.run(names, seqs, ctx=ctx, bp=bp)
t1 = time.perf_counter()
print("gpu e2e %.3f s" % (t1 - t0), got.seconds, "levels", got.levels)
its = np.array([v[0] for v in got.dd_log.values()])
print("dd iterations: sum %d median %d max %d; ncbp max %d" % (its.sum(), np.median(its), its.max(), max(v[2] for v in got.dd_log.values())))
if len(sys.argv) > 4:
    o = oracle_lib.load_oracle()
    pl = o.pipeline(names, seqs, o.params(fold_model=1), bp=bp)
    t0 = time.perf_counter(); pl.phase1(); pl.phase2(); t1 = time.perf_counter()
    print("oracle e2e %.3f s" % (t1 - t0), pl.seconds())
    print("output identical:", pl.output() == got.output)
