"""Entries per pair (p > th) on the headline set: sizes the sparse outputs (tuning aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dafs_amd import capi, synth
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 150
fam = len(sys.argv) > 3 and sys.argv[3] == "family"
recs = synth.family_set(n, L, seed=12346) if fam else synth.random_set(n, L, seed=12345)
ctx = capi.Context(0)
ctx.set_sequences([r[1] for r in recs])
res = ctx.align_posteriors(fetch=True)
nnz = np.asarray(res.nnz)
print("pairs", len(nnz), "entries/pair mean %.1f max %d" % (nnz.mean(), nnz.max()), "cells/pair ~", L * L)
