import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from dafs_amd import synth, pipeline
recs = synth.family_set(6, 50, seed=1)
res = pipeline.run([r[0] for r in recs], [r[1] for r in recs], skip_uncoupled_folds=False)
print(res.output[:200], flush=True)
