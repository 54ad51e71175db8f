"""BASELINE.json config 3 ("600 subgradient iters") on its own, for rocprofv3: every node of the N=128, L~150 run is
forced to t_max = 600 iterations (the violated == 0 exit ignored), so k_dd_solve is timed on a fixed amount of work."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dafs_amd import capi, pipeline, synth
n, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 150)
recs = synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
ctx = capi.Context(0)
pipeline.run(names, seqs, ctx=ctx)  # warm-up
r = pipeline.run(names, seqs, ctx=ctx, skip_uncoupled_folds=False, force_iters=1)
nit = int(np.sum([v[0] for v in r.dd_log.values()]))
print("forced run: %d node-iterations in %.3f s progressive (%.0f node-iterations/s)" % (nit, r.seconds["progressive"], nit / r.seconds["progressive"]))
ctx.close()
