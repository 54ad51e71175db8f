"""Does a dense base-pair input overflow the register form of the folding DP?  Prints the per-node stamp lines of a
run kept inside the leader's workgroup (DAFS_HIP_DD_SPLIT=0), whose slow-xy counters count the fallbacks (tuning aid)."""
import os, sys
os.environ["DAFS_HIP_DD_STAMPS"] = "1"; os.environ["DAFS_HIP_DD_SPLIT"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dafs_amd import synth, pipeline
from test_pct_gpu import random_bp
n, L, dens = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
recs = synth.family_set(n, L, seed=61)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
pipeline.run(names, seqs, bp=random_bp(seqs, 61, density=dens), t_max=12, level_sync=True, skip_uncoupled_folds=False)
