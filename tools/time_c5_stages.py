"""per-kernel device times of one c5 run (dafs_hip_stage_timing); usage: c5_stages.py [family|random] [N L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dafs_amd import capi, synth, pipeline
fam = len(sys.argv) > 1 and sys.argv[1] == "family"
n, L = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 400)
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
ctx = capi.Context(0)
ctx.stage_timing(True)
t = time.perf_counter()
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs], ctx=ctx, align_model=capi.ALIGN_CONTRALIGN if n >= 512 else capi.ALIGN_PROBCONS)
print("wall %.1f s" % (time.perf_counter() - t), {k: round(v, 2) for k, v in res.seconds.items()}, flush=True)
for k, (ms, longest, cnt) in sorted(ctx.stage_report().items(), key=lambda kv: -kv[1][0]):
    print("%-26s %10.1f ms  %5d launches  longest %9.1f ms" % (k, ms, cnt, longest))
