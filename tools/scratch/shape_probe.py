import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from dafs_amd import capi, pipeline, synth
rnd = synth.random_set(128, 400, seed=12345)
names, seqs = [r[0] for r in rnd], [r[1] for r in rnd]
for th_s in (0.2, 0.05, 0.03):
    got = pipeline.run(names, seqs, align_model=capi.ALIGN_CONTRALIGN, t_max=4, th_s=th_s, skip_uncoupled_folds=False)
    score, left, right = got.tree
    rows = {}
    def nrows(i):
        if i not in rows:
            rows[i] = 1 if left[i] < 0 else nrows(int(left[i])) + nrows(int(right[i]))
        return rows[i]
    print("th_s", th_s)
    for i in sorted(got.dd_dims):
        r1, r2 = nrows(int(left[i])), nrows(int(right[i]))
        if r1 + r2 >= 16:
            print(i, r1, r2, got.dd_dims[i], got.dd_log[i][:3])
