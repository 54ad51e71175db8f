"""CPU test of the N>1 path (world_size 2, gloo): pair sharding + the single all-gather.  The shard
outputs are produced by the oracle in the exact layout the pair kernel writes (including an
arbitrary, rank-dependent pool order, as the device-side bump allocator gives), gathered with
dafs_amd.dist.ShardExchange, and every pair is checked on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    from dafs_amd import dist as dd, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load_oracle()
        seqs = [s for _, s in synth.random_set(7, 30, seed=77)]
        lens = [len(s) for s in seqs]
        px, py, total = dd.shard_pairs(lens, world, rank)
        assert total == 21
        # oracle results laid out like k_pairhmm3's outputs
        rows, nnz, sims = [], [], []
        for x, y in zip(px, py):
            rp, col, val = orc.align_calculate(seqs[x], seqs[y], 0.01, 0)
            r = np.repeat(np.arange(lens[x], dtype=np.uint32), np.diff(rp))
            o = np.lexsort((r, col))
            trp = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=lens[y]))]).astype(np.uint32)
            rows.append((rp, col, val, trp, r[o], val[o]))
            nnz.append(len(col))
            sims.append(orc.similarity(rp, col, val, lens[x], lens[y]))
        rng = np.random.default_rng(100 + rank)         # arbitrary pool order (bump allocator)
        order = rng.permutation(len(px))
        off = np.zeros(len(px), np.int64)
        top = 0
        for k in order:
            off[k] = top
            top += 2 * nnz[k]
        cap = top + 17
        col = np.zeros(cap, np.uint32); val = np.zeros(cap, np.float32)
        rowptr = []
        for k, (rp, c, v, trp, tc, tv) in enumerate(rows):
            col[off[k]:off[k] + nnz[k]] = c; val[off[k]:off[k] + nnz[k]] = v
            col[off[k] + nnz[k]:off[k] + 2 * nnz[k]] = tc; val[off[k] + nnz[k]:off[k] + 2 * nnz[k]] = tv
            rowptr += [rp, trp]
        rowptr = np.concatenate(rowptr).astype(np.uint32)
        dev = torch.device("cpu")
        ex = dd.ShardExchange(dist, dev, world, len(px), len(rowptr), cap)
        ex.exchange(torch.tensor(nnz, dtype=torch.int32), torch.tensor(np.array(sims, np.float32)), torch.from_numpy(off),
                    torch.from_numpy(rowptr.view(np.int32)), torch.from_numpy(col.view(np.int32)), torch.from_numpy(val), top)
        # a repeat with the sizes of the first call (what bench.py does in its timed steps)
        ex.exchange(torch.tensor(nnz, dtype=torch.int32), torch.tensor(np.array(sims, np.float32)), torch.from_numpy(off),
                    torch.from_numpy(rowptr.view(np.int32)), torch.from_numpy(col.view(np.int32)), torch.from_numpy(val), None)
        g = ex.gathered(lens)
        n = len(seqs)
        seen = 0
        for x in range(n):
            for y in range(x + 1, n):
                rp, c, v = orc.align_calculate(seqs[x], seqs[y], 0.01, 0)
                grp, gc, gv = g.csr(x, y)
                assert np.array_equal(grp, rp) and np.array_equal(gc, c) and gv.tobytes() == v.tobytes(), (x, y)
                trp, tc, tv = g.csr(y, x)
                r = np.repeat(np.arange(lens[x], dtype=np.uint32), np.diff(rp))
                o = np.lexsort((r, c))
                assert np.array_equal(tc, r[o]) and tv.tobytes() == v[o].tobytes(), (y, x)
                assert np.float32(g.sim(x, y)).tobytes() == np.float32(orc.similarity(rp, c, v, lens[x], lens[y])).tobytes()
                seen += 1
        assert seen == total
        sm = g.sim_matrix()
        assert np.array_equal(sm, sm.T) and np.all(np.diag(sm) == 1)
        # the hand-over to Context.set_mp: every pair's forward rows, row-major pair order
        fn, frp, fc, fv = g.forward_arrays()
        r0 = e0 = p = 0
        for x in range(n):
            for y in range(x + 1, n):
                rp, c, v = orc.align_calculate(seqs[x], seqs[y], 0.01, 0)
                assert fn[p] == len(c) and np.array_equal(frp[r0:r0 + lens[x] + 1], rp)
                assert np.array_equal(fc[e0:e0 + len(c)], c) and fv[e0:e0 + len(c)].tobytes() == v.tobytes()
                r0 += lens[x] + 1; e0 += len(c); p += 1
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_shard_partition_is_exact():
    sys.path.insert(0, ROOT)
    from dafs_amd import dist as dd
    lens = [150, 141, 160, 152, 149, 158, 144, 151, 80]
    for world in (1, 2, 3, 4, 8):
        seen = set()
        sizes = []
        for r in range(world):
            px, py, total = dd.shard_pairs(lens, world, r)
            assert total == 36
            cost = np.array(lens)[px] * np.array(lens)[py]
            assert np.all(np.diff(cost) <= 0)            # longest first
            for x, y in zip(px, py):
                assert x < y and (x, y) not in seen
                seen.add((int(x), int(y)))
            sizes.append(len(px))
        assert len(seen) == 36 and max(sizes) - min(sizes) <= 1


def test_two_rank_all_gather_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _worker_gathers(rank, world, port, q):
    """the gather helpers of dist.phase1_sharded on ragged inputs (an empty contribution included)"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dafs_amd import dist as dd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        parts = [np.arange(5, dtype=np.uint32) * 3, np.zeros(0, np.uint32), np.array([7, 9], np.uint32)][:world]
        got = dd.allgather_concat(dist, parts[rank], dev)
        assert got.dtype == np.uint32 and np.array_equal(got, np.concatenate(parts))
        fparts = [np.array([0.5, -2.0], np.float32), np.array([1e-9], np.float32), np.zeros(0, np.float32)][:world]
        got = dd.allgather_concat(dist, fparts[rank], dev)
        assert got.dtype == np.float32 and got.tobytes() == np.concatenate(fparts).tobytes()
        full = np.zeros(11, np.uint32)
        lo, hi = dd.pair_ranges(11, world)[rank], dd.pair_ranges(11, world)[rank + 1]
        full[lo:hi] = np.arange(lo, hi) + 100
        s = dd.allreduce_sum(dist, full, dev)
        assert s.dtype == np.uint32 and np.array_equal(s, np.arange(11) + 100)
        b = dd.pair_ranges(21, world)
        assert b[0] == 0 and b[-1] == 21 and all(b[k] <= b[k + 1] for k in range(world))
        # gather_parts: the packed one-collective form phase1_sharded uses (tensors in, tensors out; ragged and empty parts)
        tparts = [[torch.from_numpy(parts[r].view(np.int32).copy()), torch.from_numpy(fparts[r].copy())] for r in range(world)]
        gi, gf = dd.gather_parts(dist, tparts[rank], dev)
        assert gi.dtype == torch.int32 and np.array_equal(gi.numpy().view(np.uint32), np.concatenate(parts))
        assert gf.dtype == torch.float32 and gf.numpy().tobytes() == np.concatenate(fparts).tobytes()
        ge, = dd.gather_parts(dist, [torch.zeros(0, dtype=torch.int32)], dev)   # nothing from anybody
        assert ge.numel() == 0
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ragged_gathers_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gathers, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
