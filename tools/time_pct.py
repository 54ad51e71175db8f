import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth
n, L = int(sys.argv[1]), int(sys.argv[2])
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
model = 1 if (len(sys.argv) > 4 and sys.argv[4] == "contra") else 0
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
seqs = [r[1] for r in recs]
ctx = capi.Context(0)
ctx.set_sequences(seqs)
for rep in range(int(os.environ.get("REPS", "2"))):
    t = [time.perf_counter()]
    ctx.fold_posteriors(0.01); t.append(time.perf_counter())
    ctx.align_posteriors(fetch=False, model=model) if model else ctx.align_posteriors(fetch=False); t.append(time.perf_counter())
    sim = ctx.sim(); ctx.consistency(0.25, 0.25); t.append(time.perf_counter())
    print("fold %.1f | pair %.1f | pct %.1f ms" % tuple(1e3 * (b - a) for a, b in zip(t, t[1:])), flush=True)
