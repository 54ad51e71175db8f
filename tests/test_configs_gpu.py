"""The BASELINE.json configurations that round 1 never ran under -m gpu (c2, c4, c5), on the GPU through the C ABI.

c2  N=32,  L=80 exact, ProbCons + CONTRAfold: the whole run against the oracle pipeline, bit for bit (the oracle
    needs about a second for it).
c4  N=256, L~200, ProbCons: every pair through the size-independent properties, a strided sample of pairs against
    the oracle bit for bit; the whole run (family and random set) through the run properties.
c5  N=512, L~400, CONTRAlign + CONTRAfold: the same, with the 130 816 pairs taken in pair-index shards (the shard
    interface of dafs_hip_align_posteriors, what a multi-GPU run deals to its ranks); the random set reaches
    alignments of ~11 500 columns, beyond what the node kernels used to accept.
Plus: the forms that only very wide alignments reach, forced onto a small run and checked against the oracle; the
remaining hard limits refuse cleanly and leave the context usable; the node arena hands memory back.

The oracle pipeline's PCT / DD half (oracle/pipeline.c) is a restatement that cannot be pinned to the compiled
reference in this image: whole-run equalities below are "parity unpinned" to that extent;
the pair posteriors are checked against restatements pinned to oracle/_ref."""
import os
import time

import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu


def _sets(n, length, jitter=0.07):
    rnd = synth.random_set(n, length, seed=12345, jitter=jitter)
    fam = synth.family_set(n, length, seed=12346)
    return ([r[0] for r in rnd], [r[1] for r in rnd]), ([r[0] for r in fam], [r[1] for r in fam])


def _check_pairs(oracle, res, seqs, th, model, stride, first_pair=0):
    """size-independent properties of every pair of `res` + the oracle on every stride-th pair; returns the number checked"""
    checked = 0
    for p in range(len(res)):
        x, y = int(res.pair_x[p]), int(res.pair_y[p])
        l1, l2 = len(seqs[x]), len(seqs[y])
        rp, col, val = res.csr(p)
        trp, tcol, tval = res.csr(p, transposed=True)
        assert rp[0] == 0 and rp[-1] == len(col) == len(tcol) and len(rp) == l1 + 1 and len(trp) == l2 + 1
        assert np.all(val > np.float32(th)) and np.all(val <= 1)
        assert np.all(np.diff(rp.astype(np.int64)) >= 0) and (len(col) == 0 or col.max() < l2)
        rows = np.repeat(np.arange(l1, dtype=np.uint32), np.diff(rp))
        same_row = rows[1:] == rows[:-1]
        assert np.all(col[1:][same_row] > col[:-1][same_row])           # columns ascend within a row
        order = np.lexsort((rows, col))
        assert np.array_equal(tcol, rows[order]) and tval.tobytes() == val[order].tobytes()
        assert 0 < res.sim[p] <= 1
        if (first_pair + p) % stride == 0:
            orp, ocol, oval = oracle.align_calculate(seqs[x], seqs[y], th, model)
            assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and val.tobytes() == oval.tobytes(), (x, y)
            assert np.float32(res.sim[p]).tobytes() == np.float32(oracle.similarity(orp, ocol, oval, l1, l2)).tobytes()
            checked += 1
    return checked


def _check_run(res, seqs, t_max=600):
    n = len(seqs)
    width = len(res.ss_str)
    assert len(res.rows) == n and all(len(r) == width for r in res.rows)
    assert sorted(r.replace("-", "") for r in res.rows) == sorted(seqs)      # the rows spell the inputs
    cols = np.frombuffer("".join(res.rows).encode(), np.uint8).reshape(n, width)
    assert not np.any(np.all(cols == ord("-"), axis=0))                        # no all-gap column
    depth = 0
    for ch in res.ss_str:
        assert ch in "().", ch
        depth += ch == "("
        depth -= ch == ")"
        assert depth >= 0
    assert depth == 0
    its = [v[0] for v in res.dd_log.values()]
    assert len(its) == n - 1 and max(its) <= t_max


# ---------------------------------------------------------------------------------------------- c2
def test_c2_whole_run_equals_oracle(oracle):
    """N=32, L=80 exact, ProbCons + CONTRAfold: tree, alignment rows, structure line and the per-node iteration log"""
    from dafs_amd import pipeline
    (names, seqs), _ = _sets(32, 80, jitter=0.0)
    assert all(len(s) == 80 for s in seqs)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=0))
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, vi = pl.dd_log()
    pl.close()
    got = pipeline.run(names, seqs, skip_uncoupled_folds=False)   # the solver as the reference runs it: iteration counts comparable
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    fast = pipeline.run(names, seqs)                              # the drivers' default: same output
    assert fast.output == want


def test_c2_family_whole_run_equals_oracle(oracle):
    from dafs_amd import pipeline
    _, (names, seqs) = _sets(32, 80)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=0))
    pl.phase1(); pl.phase2()
    want = pl.output()
    pl.close()
    assert pipeline.run(names, seqs).output == want


# ---------------------------------------------------------------------------------------------- c4
def test_c4_pairs_properties_and_sample(oracle):
    from dafs_amd import capi
    (names, seqs), _ = _sets(256, 200)
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(seqs)
        res = ctx.align_posteriors(capi.ALIGN_PROBCONS, 0.01)
        sim = ctx.sim()
    finally:
        ctx.close()
    assert len(res) == 256 * 255 // 2
    assert np.array_equal(sim, sim.T) and np.all(np.diag(sim) == 1) and np.all(sim > 0) and np.all(sim <= 1)
    assert _check_pairs(oracle, res, seqs, 0.01, 0, stride=251) >= 130


@pytest.mark.parametrize("which", ["random", "family"])
def test_c4_whole_run_properties(which):
    from dafs_amd import pipeline
    rnd, fam = _sets(256, 200)
    names, seqs = rnd if which == "random" else fam
    a = pipeline.run(names, seqs)
    _check_run(a, seqs)
    if which == "family":  # scheduling does not change a bit (the level-synchronous schedule is the slower one)
        b = pipeline.run(names, seqs, level_sync=True)
        assert a.output == b.output and a.dd_log == b.dd_log


def test_c4_family_whole_run_equals_oracle():
    """BASELINE config 4 (N=256, L~200, ProbCons + CONTRAfold), family set, end to end against the CPU port, bit for bit
    (output text and iteration log; parity unpinned for the dafs.cpp half).  The oracle's 32 640 pair posteriors and 256
    folds run process-parallel, the rest of it on one core."""
    import oracle_lib
    from dafs_amd import pipeline
    _, (names, seqs) = _sets(256, 200)
    t0 = time.time()
    want, (it, vi) = oracle_lib.parallel_oracle_run(names, seqs)
    print("oracle: %.1f s" % (time.time() - t0))
    got = pipeline.run(names, seqs, skip_uncoupled_folds=False)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)


# ---------------------------------------------------------------------------------------------- c5
def test_c5_contralign_pairs_in_shards(oracle):
    """130 816 CONTRAlign pairs at L~400 in eight pair-index shards; properties of all, the oracle on every 1201st"""
    from dafs_amd import capi
    (names, seqs), _ = _sets(512, 400)
    npairs = 512 * 511 // 2
    ctx = capi.Context(0)
    checked = seen = 0
    t0 = time.time()
    try:
        ctx.set_sequences(seqs)
        bounds = [npairs * k // 8 for k in range(9)]
        for k in range(8):
            res = ctx.align_posteriors(capi.ALIGN_CONTRALIGN, 0.01, pair_begin=bounds[k], pair_end=bounds[k + 1])
            assert len(res) == bounds[k + 1] - bounds[k]
            checked += _check_pairs(oracle, res, seqs, 0.01, 1, stride=1201, first_pair=bounds[k])
            seen += len(res)
            del res
    finally:
        ctx.close()
    assert seen == npairs and checked >= 100
    assert time.time() - t0 < 400


@pytest.mark.parametrize("which", ["family", "random"])
def test_c5_whole_run_properties(which):
    """CONTRAlign + CONTRAfold, consistency transforms and the progressive phase on one GPU.  The random set is the one
    DD_LMAX = 4096 used to refuse: its root alignment joins 967 and ~11 500 columns."""
    from dafs_amd import capi, pipeline
    rnd, fam = _sets(512, 400)
    names, seqs = rnd if which == "random" else fam
    t0 = time.time()
    a = pipeline.run(names, seqs, align_model=capi.ALIGN_CONTRALIGN)
    _check_run(a, seqs)
    widest = max(max(d) for d in a.dd_dims.values())
    if which == "random":
        assert widest > 4096                       # the case is what it claims to be
    _check_arena(a)
    assert a.dd_demotions == 0                     # no split node lost its folders on an undisturbed device
    assert time.time() - t0 < 300


def _node_bytes_bound(l1, l2, folds=True):
    """upper bound of a resident node's device memory from its dimensions (capi_dd.cpp nodes_open).  First block: per cell
    of the L x L matrices p, q 8 + flags 0.5 + id map 4 + lists and counters 6 = 18.5 bytes, per cell of the L1 x L2
    alignment tables 26 bytes (p, q, flags, map, lists) + its two sweep-order input copies (8 bytes per padded cell) + 512 bytes
    of traceback slots per row and panel, row arrays and padding.  Second block, only when the node folds (it has consensus pairs, or skip_uncoupled_folds is off): per cell of
    L x L the Nussinov work arrays 16 + byte codes 0.5 + up to two score copies 8 = 24.5 bytes."""
    first = (19 * (l1 * l1 + l2 * l2) + 26 * (l1 + 1) * (l2 + 1) + 8 * (l1 + 63) * (l2 + 64) + 512 * (l1 + 1) * ((l2 + 2048) // 2048)
             + 128 * (l1 + l2) + (1 << 14))
    return first + (25 * (l1 * l1 + l2 * l2) if folds else 0)


def _check_arena(res):
    """The arena's high-water mark is what the nodes open at the same time need: in a round the open nodes still hold
    their blocks while the newly ready ones get theirs (pipeline.run records each round's node set), plus 36 bytes per
    consensus base pair.  Stated from the open-node sets, not from the size of the whole tree."""
    reserved, in_use, peak = res.dd_memory
    assert in_use == 0
    cbp = {i: v[2] for i, v in res.dd_log.items()}
    folds = lambda i: cbp[i] > 0 or not res.skip_uncoupled_folds
    expected = max(sum(_node_bytes_bound(l1, l2, folds(i)) + 36 * cbp[i] + 1024 for i, l1, l2 in nodes) for _, nodes in res.rounds)
    assert peak <= expected, (peak, expected)
    assert peak >= expected / 3, (peak, expected)   # and the bound is not vacuous


# ---------------------------------------------------------------------------------------------- wide forms, limits, arena
def test_wide_alignment_forms_equal_oracle(oracle):
    """DAFS_HIP_DD_WIDE=1 pushes every node through the forms only very wide alignments reach (foldings span-ordered on
    HBM tables with no sweep-order copy; the alignment DP in panels -- of 64 columns here, one column per lane, where
    second alignments beyond 2047 columns take panels of 2048 -- with its codes in HBM slots; row pointers searched in
    HBM; one averaging row per workgroup); the run must still be the oracle's, iteration log included."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(9, 70, seed=5) + synth.random_set(3, 90, seed=6)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 5, density=0.03)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1), bp=bp)
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, _ = pl.dd_log()
    pl.close()
    os.environ["DAFS_HIP_DD_WIDE"] = "1"
    try:
        got = pipeline.run(names, seqs, bp=bp, skip_uncoupled_folds=False)
    finally:
        os.environ.pop("DAFS_HIP_DD_WIDE", None)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)


def test_wide_setup_kernels_equal_oracle(oracle, monkeypatch):
    """DAFS_HIP_DD_LISTS_WIDE=1 sends every node through the set-up kernels that only alignments of 1536+ columns take (row
    lists, envelope ends and table initialisation on grids over the rows, k_lists_rows & co.); the run -- lists, envelope,
    consensus pairs, hence every iteration -- must still be the oracle's."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(8, 90, seed=15) + synth.random_set(3, 70, seed=16)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 15, density=0.03)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1), bp=bp)
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, vi = pl.dd_log()
    pl.close()
    monkeypatch.setenv("DAFS_HIP_DD_LISTS_WIDE", "1")
    got = pipeline.run(names, seqs, bp=bp, skip_uncoupled_folds=False)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)


def test_alignment_codes_in_hbm_slots_equal_oracle(oracle, monkeypatch):
    """DAFS_HIP_DD_NWG=1 keeps the alignment DP's traceback codes out of LDS, so every node takes the register form that
    writes them to HBM, one 64-bit slot per (row, lane) (nw_wave_reg<W, 2>), and the wave traceback that reads them from
    there -- what second alignments of 1024 to 2047 columns, and narrower ones whose table does not fit LDS, run on.
    Output and iteration log must be the oracle's."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(8, 90, seed=35) + synth.random_set(3, 70, seed=36)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 45, density=0.03)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1), bp=bp)
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, vi = pl.dd_log()
    pl.close()
    monkeypatch.setenv("DAFS_HIP_DD_NWG", "1")
    got = pipeline.run(names, seqs, bp=bp, skip_uncoupled_folds=False)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)


def test_alignment_panels_of_2048_columns_equal_oracle(oracle):
    """Second alignments beyond 2047 columns: the alignment DP runs in panels of 64 x 32 columns with the panel's last column
    handed on through the edge array, the folding of that width takes the workgroup form, and every set-up kernel sees rows
    of more than 2048 cells.  The model kernels stop at 2047 nt, so matching and base-pairing rows are supplied
    (--align-aux / --fold-aux paths, on both sides): three sequences of 2 150, 2 090 and 140 nt, rows near the diagonal.
    Output and iteration log must be the oracle's (t_max bounds its time)."""
    from dafs_amd import pipeline
    rng = np.random.default_rng(91)
    lens = [2150, 2090, 140]
    seqs = ["".join(rng.choice(list("ACGU"), L)) for L in lens]
    names = ["s%d" % k for k in range(3)]
    bp = []
    for L in lens:   # a few stems per sequence: pairs (i, j) with runs of stacked neighbours
        rows = [[] for _ in range(L)]
        for _ in range(max(2, L // 60)):
            i0 = int(rng.integers(0, L - 30)); span = int(rng.integers(8, 28)); n = int(rng.integers(3, 7))
            for d in range(n):
                i, j = i0 + d, i0 + span - d
                if j - i >= 4 and j < L and not any(c == j for c, _ in rows[i]):
                    rows[i].append((j, np.float32(0.25 + 0.7 * rng.random())))
        for r in rows:
            r.sort()
        rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint32)
        bp.append((rp, np.array([c for r in rows for c, _ in r], np.uint32), np.array([v for r in rows for _, v in r], np.float32)))
    def rows_for(x, y):
        Lx, Ly = lens[x], lens[y]
        rp, col, val = [0], [], []
        for i in range(Lx):
            c0 = int(i * Ly / Lx)
            ks = sorted(set(int(k) for k in (c0 + rng.integers(-2, 3, size=2)) if 0 <= k < Ly))
            for k in ks:
                col.append(k); val.append(np.float32(0.05 + 0.4 * rng.random()))
            rp.append(len(col))
        return np.array(rp, np.uint32), np.array(col, np.uint32), np.array(val, np.float32)
    rows_of = {(x, y): rows_for(x, y) for x in range(3) for y in range(x + 1, 3)}
    order = [(x, y) for x in range(3) for y in range(x + 1, 3)]
    mp = (np.array([len(rows_of[p][1]) for p in order], np.uint32), np.concatenate([rows_of[p][0] for p in order]),
          np.concatenate([rows_of[p][1] for p in order]), np.concatenate([rows_of[p][2] for p in order]))
    kw = dict(t_max=6, th_s=0.1)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1, align_model=2, th_s1=0.1, **kw), bp=bp, mp=lambda x, y: rows_of[(x, y)])
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, vi = pl.dd_log()
    pl.close()
    got = pipeline.run(names, seqs, bp=bp, mp=mp, skip_uncoupled_folds=False, **kw)
    assert max(max(d) for d in got.dd_dims.values()) > 2047
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)


@pytest.mark.parametrize("density", [0.03, 0.3])
def test_workgroup_folding_form_equals_oracle(oracle, monkeypatch, density):
    """DAFS_HIP_DD_WG=2 gives every folder of a split node the workgroup form of the folding DP that only foldings beyond
    1024 columns take (nuss_wg_span: rolling rows and the first candidates of every column in LDS, tables by span); with
    dense base-pairing inputs the columns hold more candidates than the on-chip heads, so the global list tails run too.
    Output and iteration log must be the oracle's."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(8, 90, seed=25) + synth.random_set(3, 70, seed=26)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 35, density=density)
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1, t_max=60), bp=bp)
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, vi = pl.dd_log()
    pl.close()
    monkeypatch.setenv("DAFS_HIP_DD_WG", "2")
    got = pipeline.run(names, seqs, bp=bp, t_max=60, skip_uncoupled_folds=False)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)


def test_limits_refuse_cleanly_and_context_survives(oracle):
    """what is left of the hard limits: a pair-HMM column sequence beyond 64 lanes x 32 columns, a CONTRAfold sequence
    whose per-position tables outgrow LDS.  Each must come back as ETOOLONG (-4), and the same context must then work."""
    from dafs_amd import capi
    small = [s for _, s in synth.random_set(4, 40, seed=3)]
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(["ACGU" * 20, "ACGU" * 513])  # the second sequence of a pair spans the columns: 2052 > 2047
        with pytest.raises(capi.DafsHipError) as e:
            ctx.align_posteriors(capi.ALIGN_PROBCONS, 0.01)
        assert "code -4" in str(e.value)
        ctx.set_sequences(["ACGU" * 600, "ACGU" * 20])  # 2400 nt: 12 * (L + 2) ints do not fit the fold kernel's LDS
        with pytest.raises(capi.DafsHipError) as e:
            ctx.fold_posteriors(0.01)
        assert "code -4" in str(e.value)
        ctx.set_sequences(small)
        res = ctx.align_posteriors(capi.ALIGN_PROBCONS, 0.01)
        rp, col, val = res.csr(0)
        orp, ocol, oval = oracle.align_calculate(small[0], small[1], 0.01, 0)
        assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and val.tobytes() == oval.tobytes()
    finally:
        ctx.close()


def test_node_arena_returns_memory():
    """a long progressive run holds the open nodes, not every node it has ever solved"""
    from dafs_amd import pipeline
    recs = synth.random_set(96, 60, seed=8)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    res = pipeline.run(names, seqs)
    _check_arena(res)
    everything = sum(_node_bytes_bound(l1, l2) for l1, l2 in res.dd_dims.values())
    assert res.dd_memory[2] < everything / 2           # far below what the whole tree would take
    # nodes without consensus pairs leave their foldings out and then carry no folding arrays (the second block is carved
    # after the count): with the reference's behaviour (every node folds) the same run needs more, and says the same
    full = pipeline.run(names, seqs, skip_uncoupled_folds=False)
    _check_arena(full)
    assert full.output == res.output
    if any(v[2] == 0 for v in res.dd_log.values()):
        assert res.dd_memory[2] < full.dd_memory[2]


def test_wide_contralign_nodes_at_the_round2_fault_shape(oracle):
    """Concurrent nodes of 8 + 8 and 16 + 16 rows, ~800-1600 columns wide, over CONTRAlign rows, with consensus base pairs
    and multiplier updates -- the shape at which round 2's c5 run hit a GPU memory-access fault (foldings
    beyond 1024 columns have no register form and keep no sweep-order score copy; the multiplier updates wrote through
    the null pointer as soon as such a node's folding predicted a base pair).  128 random sequences of ~400 nt give a
    guide tree with eight 8 + 8 nodes and three 16 + 16 nodes at those widths; -t 0.05 makes their foldings predict
    pairs (at the default 0.2 the averaged matrices of unrelated sequences stay below the threshold and such nodes end
    after one pass).  Matching and base-pairing rows come from the device models (checked against the oracle elsewhere)
    and are handed to the oracle pipeline as --align-aux / --fold-aux input, so that the oracle's share is the
    consistency transforms, the tree and the progressive phase; t_max = 4 bounds its time."""
    from dafs_amd import capi, pipeline
    (names, seqs), _ = _sets(128, 400)
    n = len(seqs)
    kw = dict(t_max=4, th_s=0.05)
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(seqs)
        ctx.fold_posteriors(0.01)
        bp = [(r.copy(), c.copy(), v.copy()) for r, c, v in ctx.bp(0)]
        res = ctx.align_posteriors(capi.ALIGN_CONTRALIGN, 0.01)
        where = {(int(res.pair_x[p]), int(res.pair_y[p])): p for p in range(len(res))}
        rows_of = [res.csr(where[(x, y)]) for x in range(n) for y in range(x + 1, n)]
        mp = (np.array([len(r[1]) for r in rows_of], np.uint32), np.concatenate([r[0] for r in rows_of]),
              np.concatenate([r[1] for r in rows_of]), np.concatenate([r[2] for r in rows_of]))
        got = pipeline.run(names, seqs, ctx=ctx, bp=bp, mp=mp, skip_uncoupled_folds=False, **kw)   # resident nodes, in rounds
        score, left, right = got.tree
        rows = {}
        def nrows(i):
            if i not in rows:
                rows[i] = 1 if left[i] < 0 else nrows(int(left[i])) + nrows(int(right[i]))
            return rows[i]
        info = {i: (nrows(int(left[i])), nrows(int(right[i])), got.dd_dims[i], got.dd_log[i][:3]) for i in got.dd_dims}
        wide = "\n".join("%d: rows %d+%d columns %s (iterations, violated, ncbp) %s" % ((i,) + info[i]) for i in sorted(info) if sum(info[i][:2]) >= 16)
        if os.path.isdir("gpurun_out"):
            open("gpurun_out/r3_fault_shape_nodes.txt", "w").write(wide + "\n")
        # the case is what it claims to be: the node shapes, and on them what faulted -- multiplier updates (consensus pairs,
        # violated constraints, a second iteration) on nodes whose foldings have no register form (beyond 1024 columns)
        busy = {i for i, (r1, r2, d, (its, viol, ncbp)) in info.items() if ncbp > 0 and its > 1 and viol > 0}
        assert len([i for i in busy if info[i][:2] == (8, 8) and 700 <= min(info[i][2]) and max(info[i][2]) <= 1150]) >= 6, wide
        assert len([i for i in busy if info[i][:2] == (16, 16) and min(info[i][2]) > 1024]) >= 3, wide
        _check_arena(got)
        t0 = time.time()
        pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1, align_model=2, th_s1=kw["th_s"], **kw), bp=bp,
                             mp=lambda x, y: res.csr(where[(x, y)]))
        pl.phase1(); pl.phase2()
        want = pl.output()
        it, vi = pl.dd_log()
        secs = pl.seconds()
        pl.close()
        print("oracle: %.1f s (pct+tree %.1f, progressive %.1f)" % (time.time() - t0, secs[2], secs[3]))
        assert got.output == want
        assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
        assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)
        lvl = pipeline.run(names, seqs, ctx=ctx, bp=bp, mp=mp, level_sync=True, skip_uncoupled_folds=False, **kw)  # the nodes of a level in ONE launch
        assert lvl.output == want and lvl.dd_log == got.dd_log
    finally:
        ctx.close()


@pytest.mark.parametrize("stage", [1, 2, 3, 4])
def test_failed_open_leaves_nothing_behind(stage):
    """A dafs_hip_nodes_open that fails late (DAFS_HIP_DD_FAIL_OPEN injects DAFS_HIP_ELAUNCH after the blocks and their
    fills are queued / after the lists / after the second blocks / at the very end) gives every block back once the
    stream has drained: the arena's in-use figure returns to where it was, and the context then runs the same nodes."""
    from dafs_amd import capi, pipeline
    recs = synth.family_set(8, 60, seed=13)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    want = pipeline.run(names, seqs)
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(seqs)
        ctx.fold_posteriors(0.01)
        ctx.align_posteriors(capi.ALIGN_PROBCONS, 0.01, fetch=False)
        ctx.consistency(0.25, 0.25)
        prm = capi.dd_params()
        one = lambda i: (np.array([i], np.uint32), np.ones((1, len(seqs[i])), np.uint8))
        keep, _ = ctx.nodes_open([one(0) + one(1)], prm)          # an open node that must survive the failed call
        before = ctx.nodes_memory()[1]
        assert before > 0
        os.environ["DAFS_HIP_DD_FAIL_OPEN"] = str(stage)
        try:
            with pytest.raises(capi.DafsHipError):
                ctx.nodes_open([one(2) + one(3), one(4) + one(5)], prm)
        finally:
            os.environ.pop("DAFS_HIP_DD_FAIL_OPEN", None)
        assert ctx.nodes_memory()[1] == before
        hs, dims = ctx.nodes_open([one(2) + one(3)], prm)
        assert hs == [keep[0] + 1]                                 # the failed call's handles were not consumed
        assert all(ctx.nodes_advance(keep + hs, prm, 0))
        for h, i, j in ((keep[0], 0, 1), (hs[0], 2, 3)):
            o = ctx.nodes_result(h, len(seqs[i]), len(seqs[j]))
            assert o["iterations"] >= 1
        assert ctx.nodes_memory()[1] == 0
        ctx.nodes_close()
        again = pipeline.run(names, seqs, ctx=ctx)
        assert again.output == want.output and again.dd_log == want.dd_log
    finally:
        ctx.close()
