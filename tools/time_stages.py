"""Wall-clock of each device stage on the headline set (after a warm-up), for tuning."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dafs_amd import capi, synth, pipeline
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 150
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
ctx = capi.Context(0)
for rep in range(2):
    t = [time.perf_counter()]
    ctx.set_sequences(seqs); t.append(time.perf_counter())
    ctx.fold_posteriors(0.01); t.append(time.perf_counter())
    ctx.align_posteriors(fetch=False); t.append(time.perf_counter())
    sim = ctx.sim(); ctx.consistency(0.25, 0.25); t.append(time.perf_counter())
    tree = capi.build_tree(sim); t.append(time.perf_counter())
    if rep:
        print("set_seq %.1f ms | fold %.1f | pair(L1 call) %.1f | pct %.1f | tree %.1f" % tuple(1e3 * (b - a) for a, b in zip(t, t[1:])))
res = pipeline.run(names, seqs, ctx=ctx, level_sync=os.environ.get("DAFS_LEVEL_SYNC") == "1", slice_iters=(int(os.environ["DAFS_SLICE"]) if "DAFS_SLICE" in os.environ else None),
                   skip_uncoupled_folds=os.environ.get("DAFS_SKIP", "1") != "0")
print({k: round(v, 3) for k, v in res.seconds.items()}, "levels", res.levels, "cols", len(res.rows[0]))
its = sorted(v[0] for v in res.dd_log.values())
print("dd its sum", sum(its), "top", its[-8:])
