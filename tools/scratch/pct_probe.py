import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from dafs_amd import capi, synth
orc = oracle_lib.load_oracle()
recs = synth.random_set(4, 30, seed=98)
names, seqs = [r[0] for r in recs], [r[1] for r in recs]
pl = orc.pipeline(names, seqs, orc.params(fold_model=0, w_pct_s=0.0)); pl.phase1()
ctx = capi.Context(0); ctx.set_sequences(seqs); ctx.fold_posteriors(0.01); ctx.align_posteriors(fetch=False); ctx.sim(); ctx.consistency(0.25, 0.0)
mp = ctx.mp(1)
rp, col, val = pl.mp(0, 1); grp, gcol, gval = mp.csr(0)
for i in range(6):
    print("row", i, "want", list(zip(col[rp[i]:rp[i+1]], np.round(val[rp[i]:rp[i+1]], 4))))
    print("      got ", list(zip(gcol[grp[i]:grp[i+1]], np.round(gval[grp[i]:grp[i+1]], 4))))
