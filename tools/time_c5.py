import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth, pipeline
fam = len(sys.argv) > 1 and sys.argv[1] == "family"
recs = synth.family_set(512, 400, seed=12346) if fam else synth.random_set(512, 400, seed=12345)
t = time.perf_counter()
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs], align_model=capi.ALIGN_CONTRALIGN)
print("c5", "family" if fam else "random", "wall %.1f s" % (time.perf_counter() - t), {k: round(v, 2) for k, v in res.seconds.items()}, "cols", len(res.rows[0]), flush=True)
