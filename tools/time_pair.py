"""Time of the all-pairs alignment-posterior stage (L1 call, results left on the device); tuning aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
model = capi.ALIGN_CONTRALIGN if len(sys.argv) > 4 and sys.argv[4] == "contralign" else capi.ALIGN_PROBCONS
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
ctx = capi.Context(0)
ctx.set_sequences([r[1] for r in recs])
ts = []
for _ in range(6):
    t0 = time.perf_counter(); ctx.align_posteriors(model=model, fetch=False); ts.append(time.perf_counter() - t0)
print(sys.argv[1:], "align_posteriors ms: min %.3f median %.3f" % (1e3 * min(ts[1:]), 1e3 * sorted(ts[1:])[2]))
