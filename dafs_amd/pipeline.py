"""Host-side driver of the whole DAFS run on top of the C ABI -- the Python mirror of
DAFS::run (reference src/dafs.cpp:1781-1889) used by tests and bench.py.  Everything numeric is
done by libdafs_hip.so on the GPU; this file only holds the guide tree, the alignment bookkeeping
(project_alignment) and the output format.  The C++ `dafs` executable (dafs_amd/csrc/cli_main.cpp)
is the same logic for the drop-in command line."""
import heapq
import os
import sys

import numpy as np

from . import capi

NONE = 0xFFFFFFFF


def build_tree(sim):
    """DAFS::build_tree, src/dafs.cpp:446-492.  Returns (score[2n-1], left, right) with -1 for leaves."""
    n = sim.shape[0]
    T = 2 * n - 1
    score = np.zeros(T, np.float32)
    left = -np.ones(T, np.int64)
    right = -np.ones(T, np.int64)
    d = np.zeros((n, n), np.float32)
    idx = [-1] * T
    for i in range(n):
        idx[i] = i
    pq = []
    for i in range(n - 1):
        for j in range(i + 1, n):
            d[i, j] = d[j, i] = sim[i, j]
            heapq.heappush(pq, (-float(sim[i, j]), -i, -j))  # max-heap on (sim, (i, j))
    cur = n
    while pq:
        s, a, b = heapq.heappop(pq)
        s, a, b = np.float32(-s), -a, -b
        if idx[a] != -1 and idx[b] != -1:
            l, r = idx[a], idx[b]
            idx[a] = idx[b] = -1
            for i in range(cur):
                if idx[i] != -1:
                    ii = idx[i]
                    v = np.float32(np.float32(d[ii, l] + d[ii, r]) * s) / np.float32(2)
                    d[ii, l] = d[l, ii] = v
                    heapq.heappush(pq, (-float(v), -i, -cur))
            score[cur] = s
            left[cur], right[cur] = a, b
            idx[cur] = l
            cur += 1
    return score, left, right


def tree_string(score, left, right, names, i=None):
    """print_tree, src/dafs.cpp:495-511 (operator<<(float) == %g)"""
    if i is None:
        i = len(score) - 1
    if left[i] < 0:
        return names[i]
    return "[ %g %s %s ]" % (float(score[i]), tree_string(score, left, right, names, left[i]),
                            tree_string(score, left, right, names, right[i]))


def project_alignment(a1, a2, z):
    """src/dafs.cpp:766-825.  a = (seq_idx[n], mask[n, L])"""
    s1, m1 = a1
    s2, m2 = a2
    L1, L2 = m1.shape[1], m2.shape[1]
    cols1, cols2 = [], []  # per output column: source column in aln1 / aln2 or -1
    k = 0
    for i in range(L1):
        if z[i] != NONE:
            while k < z[i]:
                cols1.append(-1); cols2.append(k); k += 1
            cols1.append(i); cols2.append(k); k += 1
        else:
            cols1.append(i); cols2.append(-1)
    while k < L2:
        cols1.append(-1); cols2.append(k); k += 1
    c1 = np.array(cols1); c2 = np.array(cols2)
    o1 = np.where(c1[None, :] >= 0, m1[:, np.maximum(c1, 0)], 0).astype(np.uint8)
    o2 = np.where(c2[None, :] >= 0, m2[:, np.maximum(c2, 0)], 0).astype(np.uint8)
    return np.concatenate([s1, s2]), np.concatenate([o1, o2], axis=0)


class Result:
    pass


def run(names, seqs, ctx=None, bp=None, w=4.0, eta0=0.5, t_max=600, w_pct_a=0.25, w_pct_s=0.25, th_a=0.01,
        th_s=0.2, th_s1=None, align_model=capi.ALIGN_PROBCONS, force_iters=0, timers=None, level_sync=False, slice_iters=None,
        mp=None, skip_uncoupled_folds=True, shard=None, round_us=None, w_pct_f=0.0, bp_update=False, bp_update1=False):
    """The whole run.  bp: per-sequence (rowptr, col, val) base-pairing rows (--fold-aux); None
    computes them with the device fold model.  mp: supplied matching probabilities (--align-aux), see Context.set_mp.
    shard: (torch.distributed module, torch device) of an initialised process group -- phase 1 (folds, pair posteriors,
    matching consistency transform) is then split over the ranks and gathered (dist.phase1_sharded); every rank
    finishes the run and holds the same result."""
    import time
    # combinations this driver does not implement are refused, not ignored (the command line, cli_main.cpp, covers
    # --bp-update with level batches through its own solve_batch)
    if level_sync and bp_update:
        raise ValueError("pipeline.run: bp_update needs the resident-node schedule (level_sync=False)")
    if shard is not None and (bp is not None or mp is not None or w_pct_f != 0.0):
        raise ValueError("pipeline.run: a sharded phase 1 computes its posteriors itself: bp / mp (--fold-aux / --align-aux) and "
                         "w_pct_f (-f) are single-process options")
    own = ctx is None
    if own:
        ctx = capi.Context(0)
    t = [time.perf_counter()]
    n = len(seqs)
    if shard is not None:
        from . import dist as ddist
        ddist.phase1_sharded(ctx, seqs, shard[0], shard[1], align_model, th_a, w_pct_a, w_pct_s)
        t += [time.perf_counter()] * 2
        sim = ctx.sim()
    else:
        sim = _phase1_local(ctx, seqs, bp, mp, align_model, th_a, w_pct_a, w_pct_s, t, w_pct_f)
    score, left, right = capi.build_tree(sim)  # same code as the command line (build_tree below is its Python twin, kept for the CPU tests)
    t.append(time.perf_counter())
    return _phase2(ctx, own, names, seqs, n, sim, score, left, right, t, w, eta0, t_max, th_a, th_s, th_s1, force_iters, level_sync, slice_iters,
                   skip_uncoupled_folds, round_us, bp_update, bp_update1)


def _phase1_local(ctx, seqs, bp, mp, align_model, th_a, w_pct_a, w_pct_s, t, w_pct_f=0.0):
    import time
    ctx.set_sequences(seqs)
    # The folding (one workgroup per sequence) leaves most of the device idle, and nothing before the base-pair
    # transform needs its result: it is started on its own stream, and the all-pairs alignment posteriors and the
    # matching-probability transform run beside it.
    if bp is not None:
        ctx.set_bp(bp)
    else:
        ctx.fold_begin(0.01)
    t.append(time.perf_counter())
    if mp is not None:
        ctx.set_mp(*mp)  # (nnz, rowptr, col, val) of every pair, --align-aux
    else:
        ctx.align_posteriors(align_model, th_a, fetch=False)
    t.append(time.perf_counter())
    folded = bp is not None
    if w_pct_f != 0.0:  # relax_fourway_consistency (dafs.cpp:1808): needs the base-pairing rows, replaces mp_ before sim_
        if not folded:
            ctx.fold_end()
            folded = True
        ctx.fourway_consistency(w_pct_f)
    sim = ctx.sim()
    ctx.consistency_match(w_pct_a)
    if not folded:
        ctx.fold_end()
    ctx.consistency_bp(w_pct_s)
    return sim


def _phase2(ctx, own, names, seqs, n, sim, score, left, right, t, w, eta0, t_max, th_a, th_s, th_s1, force_iters, level_sync, slice_iters,
            skip_uncoupled_folds, round_us=None, bp_update=False, bp_update1=False):
    import time
    res = Result()
    res.sim = sim
    res.tree = (score, left, right)
    res.tree_line = tree_string(score, left, right, names)
    # progressive phase.  level_sync: solve every node whose children are ready, level by level (one blocking
    # call per level).  Otherwise the nodes stay resident on the device and every round advances all open
    # nodes by at most `slice_iters` iterations in one launch: a node that needs 600 iterations no longer
    # holds back the parents of its level-mates.  Same results either way.
    lens = [len(s) for s in seqs]
    aln = {i: (np.array([i], np.uint32), np.ones((1, lens[i]), np.uint8)) for i in range(n)}
    pending = [i for i in range(n, 2 * n - 1)]
    # only the alignment z of a node is consumed here (DAFS::align_alignments, dafs.cpp:896-912), so nodes that have no
    # consensus base pair to couple their subproblems need not run their two folding DPs (dafs_dd_params doc)
    prm = capi.dd_params(w=w, eta0=eta0, th_a=th_a, th_s=th_s, t_max=t_max, force_iters=force_iters,
                         skip_uncoupled_folds=1 if skip_uncoupled_folds else 0)
    trace = os.environ.get("DAFS_PIPELINE_TRACE") == "1"  # node shapes on stderr as they are opened (diagnostics)
    res.dd_log = {}
    res.dd_dims = {}  # node -> (columns of the left, of the right alignment); resident-node mode only
    res.levels = 0
    res.rounds = []  # resident-node mode: (seconds, [(node, columns left, columns right), ...]) per round (diagnostics)
    if level_sync:
        while pending:
            ready = [i for i in pending if left[i] in aln and right[i] in aln]
            outs = ctx.solve_nodes([(aln[left[i]][0], aln[left[i]][1], aln[right[i]][0], aln[right[i]][1]) for i in ready], prm)
            for i, o in zip(ready, outs):
                aln[i] = project_alignment(aln[left[i]], aln[right[i]], o["z"])
                res.dd_log[i] = (o["iterations"], o["violated"], o["ncbp"], o["score"])
                del aln[left[i]], aln[right[i]]
            pending = [i for i in pending if i not in ready]
            res.levels += 1
    else:
        # A round = one call: the open nodes advance while the nodes whose children finished in the last round are set up
        # and started beside them (Context.nodes_round).  A round ends after `round_us` microseconds (all its nodes stop at
        # the next iteration end) or, when slice_iters is given, after that many iterations of every node (tests cut the
        # loop in many ways: the results do not depend on it).
        if slice_iters is None and round_us is None:
            round_us = int(os.environ.get("DAFS_ROUND_US", "2500"))
        open_nodes = {}  # node -> (handle, len1, len2)
        while pending or open_nodes:
            ready = [i for i in pending if left[i] in aln and right[i] in aln]
            if ready and trace:
                print("open", [(i, aln[left[i]][1].shape, aln[right[i]][1].shape) for i in ready], file=sys.stderr, flush=True)
            pending = [i for i in pending if i not in ready]
            ids = sorted(open_nodes)
            t_round = time.perf_counter()
            def node_input(i):
                a1, a2 = aln[left[i]], aln[right[i]]
                if bp_update and i == 2 * n - 2:
                    # --bp-update: the top call of the recursion (DAFS::align(ss, aln, root), dafs.cpp:1518-1537) re-estimates
                    # both base-pairing matrices under the structure decoded from their averages (:919-934)
                    upd = []
                    for s_idx, msk in (a1, a2):
                        _, ss0, _ = ctx.consensus_structure(s_idx, msk, th_s)
                        upd.append(ctx.update_basepairing(s_idx, msk, ss0))
                    return (a1[0], a1[1], a2[0], a2[1], upd[0], upd[1])
                return (a1[0], a1[1], a2[0], a2[1])
            hs, dims, fin_old, fin_new = ctx.nodes_round([node_input(i) for i in ready],
                                                         [open_nodes[i][0] for i in ids], prm, slice_iters or 0, round_us or 0)
            for i, h, d in zip(ready, hs, dims):
                open_nodes[i] = (h, d[0], d[1])
            res.rounds.append((time.perf_counter() - t_round, [(i, open_nodes[i][1], open_nodes[i][2]) for i in ids + ready]))
            for i, f in list(zip(ids, fin_old)) + list(zip(ready, fin_new)):
                if not f:
                    continue
                h, l1, l2 = open_nodes.pop(i)
                o = ctx.nodes_result(h, l1, l2)
                aln[i] = project_alignment(aln[left[i]], aln[right[i]], o["z"])
                res.dd_log[i] = (o["iterations"], o["violated"], o["ncbp"], o["score"])
                res.dd_dims[i] = (l1, l2)
                del aln[left[i]], aln[right[i]]
            res.levels += 1
        res.dd_memory = ctx.nodes_memory()  # (reserved, in use, peak) bytes of the resident nodes
        res.dd_demotions = ctx.nodes_demotions()  # split nodes that lost their folders (0 on an undisturbed device)
        res.skip_uncoupled_folds = bool(skip_uncoupled_folds)  # nodes without consensus pairs then carry no folding arrays
        ctx.nodes_close()
    root = 2 * n - 2
    sidx, mask = aln[root]
    t.append(time.perf_counter())
    th1 = th_s if th_s1 is None else th_s1
    _, ss, _ = ctx.consensus_structure(sidx, mask, th1)
    if bp_update1:  # :1863-1869: decode, re-estimate under that structure, decode again
        _, ss = ctx.nussinov(ctx.update_basepairing(sidx, mask, ss), None, th1)
    res.ss = ss
    res.ss_str = capi.make_brackets(ss)
    order = np.argsort(sidx, kind="stable")  # std::sort(aln) :1876
    lines = [res.tree_line, ">SS_cons", res.ss_str]
    res.rows = []
    for r in order:
        row_bytes = np.full(mask.shape[1], ord("-"), np.uint8)
        row_bytes[mask[r].astype(bool)] = np.frombuffer(seqs[sidx[r]].encode("latin-1"), np.uint8)  # residues into their columns
        row = row_bytes.tobytes().decode("latin-1")
        res.rows.append(row)
        lines += ["> " + names[sidx[r]], row]
    res.output = "\n".join(lines) + "\n"
    t.append(time.perf_counter())
    # fold_launch: the folding is only started there; its kernels overlap `pair` and the first half of `pct_fold_tree`,
    # which also holds the wait for them
    res.seconds = dict(fold_launch=t[1] - t[0], pair=t[2] - t[1], pct_fold_tree=t[3] - t[2], progressive=t[4] - t[3], final=t[5] - t[4],
                       total=t[5] - t[0])
    if own:
        ctx.close()
    return res
