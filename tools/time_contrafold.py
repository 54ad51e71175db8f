import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth
n, L = int(sys.argv[1]), int(sys.argv[2])
recs = synth.random_set(n, L, seed=12345)
seqs = [r[1] for r in recs]
ctx = capi.Context(0)
ctx.set_sequences(seqs)
for rep in range(3):
    t = time.perf_counter(); ctx.fold_posteriors(0.01); print("fold ms %.2f" % (1e3 * (time.perf_counter() - t)), flush=True)
