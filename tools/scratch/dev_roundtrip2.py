"""two emulated ranks in one process: the device path of dist.phase1_sharded with torch.cat in place of the all-gather (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from dafs_amd import capi, synth, dist as dd
def say(*a): print(*a, flush=True)
dev = torch.device("cuda", 0)
recs = synth.family_set(7, 60, seed=41) + synth.random_set(4, 50, seed=42)
seqs = [r[1] for r in recs]; n = len(seqs); npairs = n * (n - 1) // 2; world = 2
i32 = lambda k: torch.empty(max(int(k), 1), dtype=torch.int32, device=dev)
f32 = lambda k: torch.empty(max(int(k), 1), dtype=torch.float32, device=dev)
ctxs = [capi.Context(0) for _ in range(world)]
for c in ctxs: c.set_sequences(seqs)
parts = []
for rank in range(world):
    mine = list(range(rank, n, world))
    fc = capi.Context(0); fc.set_sequences([seqs[x] for x in mine]); fc.fold_posteriors(0.01)
    ne, nr = fc.bp_sizes(0)
    rp, col, val = i32(nr), i32(ne), f32(ne)
    nr, ne = fc.bp_export_dev(rp.data_ptr(), col.data_ptr(), val.data_ptr(), ne); torch.cuda.synchronize(); fc.close()
    parts.append((rp[:nr], col[:ne], val[:ne])); say("rank", rank, "fold export", nr, ne)
g = [torch.cat([p[k] for p in parts]) for k in range(3)]; torch.cuda.synchronize()
order = [x for r in range(world) for x in range(r, n, world)]
for c in ctxs:
    c.set_bp_dev(order, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), int(g[1].numel())); torch.cuda.synchronize()
say("set_bp_dev done")
ref = capi.Context(0); ref.set_sequences(seqs); ref.fold_posteriors(0.01)
say("bp equal", all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2].tobytes() == y[2].tobytes() for x, y in zip(ref.bp(0), ctxs[1].bp(0))))
b = dd.pair_ranges(npairs, world)
parts = []
for rank in range(world):
    c = ctxs[rank]; cnt = b[rank + 1] - b[rank]
    c.align_posteriors(0, 0.01, pair_begin=b[rank], pair_end=b[rank + 1], fetch=False)
    _, ne, nr = c.mp_sizes(0)
    nnz, rp, col, val, sim = i32(cnt), i32(nr), i32(ne), f32(ne), f32(cnt)
    nr, ne = c.mp_export_dev(0, 0, cnt, nnz.data_ptr(), rp.data_ptr(), col.data_ptr(), val.data_ptr(), sim.data_ptr(), ne); torch.cuda.synchronize()
    parts.append((nnz[:cnt], rp[:nr], col[:ne], val[:ne], sim[:cnt])); say("rank", rank, "pairs export", cnt, nr, ne)
g = [torch.cat([p[k] for p in parts]) for k in range(5)]; torch.cuda.synchronize()
for c in ctxs:
    c.mp_install_dev(0, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), g[4].data_ptr(), int(g[2].numel())); torch.cuda.synchronize()
say("mp install done")
ref.align_posteriors(0, 0.01, fetch=False)
pa, pb = ref.mp(0), ctxs[0].mp(0)
say("mp equal", all(np.array_equal(pa.csr(p, t)[k], pb.csr(p, t)[k]) for p in range(npairs) for t in (False, True) for k in range(2)) and np.array_equal(ref.sim(), ctxs[0].sim()))
parts = []
for rank in range(world):
    c = ctxs[rank]; cnt = b[rank + 1] - b[rank]
    c.consistency_match_range(0.25, b[rank], b[rank + 1]); say("rank", rank, "pct range done")
    _, ne, nr = c.mp_sizes(1)
    nnz, rp, col, val = i32(cnt), i32(nr), i32(ne), f32(ne)
    nr, ne = c.mp_export_dev(1, b[rank], cnt, nnz.data_ptr(), rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, ne); torch.cuda.synchronize()
    parts.append((nnz[:cnt], rp[:nr], col[:ne], val[:ne])); say("rank", rank, "relaxed export", nr, ne)
g = [torch.cat([p[k] for p in parts]) for k in range(4)]; torch.cuda.synchronize()
for c in ctxs:
    c.mp_install_dev(1, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), None, int(g[2].numel())); torch.cuda.synchronize()
say("relaxed install done")
ref.consistency_match(0.25)
pa, pb = ref.mp(1), ctxs[1].mp(1)
say("relaxed equal", all(np.array_equal(pa.csr(p, t)[k], pb.csr(p, t)[k]) for p in range(npairs) for t in (False, True) for k in range(2)))
for c in ctxs: c.consistency_bp(0.25)
say("bp pct done")
