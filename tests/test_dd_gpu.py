"""GPU parity for the progressive phase: decoders against the golden
vectors and the oracle; averaging + solve_by_dd + projection (a19, a20, a23) and the whole run
against the oracle pipeline and the known answers."""
import os

import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from dafs_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_decoders_golden(ctx):
    z = np.load(os.path.join(G, "decoders.npz"))
    for k in range(int(z["n"])):
        p, q, w, th = z["p%d" % k], z["q%d" % k], float(z["w%d" % k]), float(z["th%d" % k])
        s, ss = ctx.nussinov(p, q, th, w)
        assert np.float32(s).tobytes() == np.float32(z["s%d" % k]).tobytes() and np.array_equal(ss, z["ss%d" % k]), k
        s, ss = ctx.nussinov(p, None, th)
        assert np.float32(s).tobytes() == np.float32(z["sf%d" % k]).tobytes() and np.array_equal(ss, z["ssf%d" % k]), k
        from dafs_amd import capi
        assert capi.make_brackets(ss) == str(z["br%d" % k])
        pz, qz, tha = z["pz%d" % k], z["qz%d" % k], float(z["tha%d" % k])
        s, al = ctx.nw(pz, qz, tha)
        assert np.float32(s).tobytes() == np.float32(z["sz%d" % k]).tobytes() and np.array_equal(al, z["al%d" % k]), k
        s, al = ctx.nw(pz, None, tha)
        assert np.float32(s).tobytes() == np.float32(z["szf%d" % k]).tobytes() and np.array_equal(al, z["alf%d" % k]), k


def test_decoders_fuzz_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(77)
    for t in range(25):
        L = int(rng.choice([1, 2, 3, 4, 7, 33, 90, 200]))
        L2 = int(rng.choice([1, 2, 5, 31, 100, 170]))
        dens = float(rng.choice([0.0, 0.03, 0.2]))
        p = (rng.random((L, L)) * (rng.random((L, L)) < dens)).astype(np.float32)
        if t % 2:
            p = (np.round(p * 4) / 4).astype(np.float32)
        q = ((rng.random((L, L)) - 0.4) * (rng.random((L, L)) < 0.3)).astype(np.float32)
        w, th = float(rng.choice([4.0, 1.3333334])), float(rng.choice([0.2, 0.05]))
        a, b = ctx.nussinov(p, q, th, w), oracle.nussinov(p, q, th, w)
        assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), (t, L)
        pz = (rng.random((L, L2)) * (rng.random((L, L2)) < max(dens, 0.02))).astype(np.float32)
        qz = (rng.random((L, L2)) * (rng.random((L, L2)) < 0.2)).astype(np.float32)
        env = ctx.nw_envelope(pz, 0.01)
        assert np.array_equal(env, oracle.nw_envelope(pz, 0.01)), (t, L, L2)
        a, b = ctx.nw(pz, qz, 0.01, env), oracle.nw(pz, qz, 0.01)
        assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), (t, L, L2)


def test_nussinov_decoder_forms_vs_oracle(ctx, oracle, monkeypatch):
    """The standalone folding decoder takes the workgroup form (rolling rows + candidate heads in LDS) up to ~9 900 columns and
    the span-ordered form on global tables beyond; DAFS_HIP_NUSS_GLOBAL=1 keeps the latter at any width.  Both against the
    oracle, with quantised inputs (ties) and dense ones (more candidates per column than the on-chip heads)."""
    rng = np.random.default_rng(78)
    cases = []
    for t in range(12):
        L = int(rng.choice([3, 5, 64, 65, 130, 257, 600]))
        dens = float(rng.choice([0.02, 0.2, 0.6]))
        p = (rng.random((L, L)) * (rng.random((L, L)) < dens)).astype(np.float32)
        if t % 2:
            p = (np.round(p * 4) / 4).astype(np.float32)
        q = ((rng.random((L, L)) - 0.4) * (rng.random((L, L)) < 0.3)).astype(np.float32) if t % 3 else None
        w, th = float(rng.choice([4.0, 1.3333334])), float(rng.choice([0.2, 0.05]))
        cases.append((p, q, th, w, oracle.nussinov(p, q, th, w)))
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("DAFS_HIP_NUSS_GLOBAL", env)
        for k, (p, q, th, w, b) in enumerate(cases):
            a = ctx.nussinov(p, q, th, w)
            assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), (env, k, p.shape[0])


def _run_both(oracle, names, seqs, bp, **kw):
    from dafs_amd import pipeline
    okw = dict(fold_model=1)
    for k in ("w", "eta0", "t_max", "w_pct_a", "w_pct_s", "th_a", "th_s"):
        if k in kw:
            okw[k] = kw[k]
    if "th_s" in kw:
        okw["th_s1"] = kw["th_s"]
    pl = oracle.pipeline(names, seqs, oracle.params(**okw), bp=bp)
    pl.phase1(); pl.phase2()
    want = pl.output()
    it, vi = pl.dd_log()
    pl.close()
    kw.setdefault("skip_uncoupled_folds", False)  # these tests compare the iteration log with the oracle's: the solver as the reference runs it
    got = pipeline.run(names, seqs, bp=bp, **kw)
    return want, (it, vi), got


def test_rf00005_whole_run(oracle):
    import test_oracle_cpu as t
    ka = t.known()
    recs = oracle.fasta(os.path.join(G, "RF00005_0.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    want, (it, vi), got = _run_both(oracle, names, seqs, t._golden_bp(seqs))
    assert got.tree_line == ka["rf00005.probcons.tree"]
    assert got.rows[0] == ka["rf00005.probcons.contrafold.first_row"]
    assert got.rows[-1] == ka["rf00005.probcons.contrafold.last_row"]
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)


@pytest.mark.parametrize("n,length,fam,seed", [(6, 50, True, 1), (8, 90, False, 2), (5, 140, True, 3), (2, 30, False, 4)])
def test_synthetic_whole_run(oracle, n, length, fam, seed):
    from test_pct_gpu import random_bp
    recs = synth.family_set(n, length, seed=seed) if fam else synth.random_set(n, length, seed=seed)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    want, (it, vi), got = _run_both(oracle, names, seqs, random_bp(seqs, seed, density=0.02))
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)


def test_iteration_cap_and_params(oracle):
    from test_pct_gpu import random_bp
    recs = synth.family_set(6, 70, seed=9)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 9, density=0.05)
    for kw in (dict(t_max=3), dict(w=2.0, eta0=0.25, th_s=0.1), dict(t_max=1, w_pct_a=0.0)):
        want, (it, vi), got = _run_both(oracle, names, seqs, bp, **kw)
        assert got.output == want, kw


@pytest.mark.parametrize("slice_iters", [1, 7, 64])
def test_resident_nodes_equal_level_batches(oracle, slice_iters):
    """dafs_hip_nodes_open/_advance/_result: however the iterations of a node are cut into launches, and
    whichever nodes share a launch, the run is the level-synchronous one (and the oracle's)."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(12, 60, seed=21)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 21, density=0.04)
    want, (it, vi), got = _run_both(oracle, names, seqs, bp, slice_iters=slice_iters)
    ref = pipeline.run(names, seqs, bp=bp, level_sync=True, skip_uncoupled_folds=False)
    assert got.output == want == ref.output
    assert got.dd_log == ref.dd_log
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)


def test_alignments_beyond_1024_columns(oracle):
    """Child alignments longer than 64 lanes x 16 columns take the wide forms of the node kernels
    (inputs fetched at the end of the step, accumulator rows sized at launch)."""
    from test_pct_gpu import random_bp
    recs = synth.random_set(3, 1100, seed=31)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    assert max(len(s) for s in seqs) > 1024
    want, (it, vi), got = _run_both(oracle, names, seqs, random_bp(seqs, 31, density=0.003), t_max=4)
    assert got.output == want
    assert len(got.rows[0]) > 1024


def test_split_mode_and_shared_region(oracle, monkeypatch):
    """Alignments of ~190-400 columns: the folding DPs either share one LDS region (x, then y) or, when the
    launch is small, run on workgroups of their own next to the leader.  Both must reproduce the oracle."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(5, 240, seed=41)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 41, density=0.01)
    want, (it, vi), got = _run_both(oracle, names, seqs, bp, t_max=40)
    assert got.output == want
    monkeypatch.setenv("DAFS_HIP_DD_SPLIT", "0")  # the same run with the foldings kept inside the leader's workgroup
    ref = pipeline.run(names, seqs, bp=bp, t_max=40, level_sync=True, skip_uncoupled_folds=False)
    assert ref.output == want and ref.dd_log == got.dd_log


@pytest.mark.parametrize("n,length,density,t_max", [(6, 120, 0.02, 60), (5, 185, 0.015, 40), (4, 150, 0.06, 12)])
def test_span_form_placements(oracle, monkeypatch, n, length, density, t_max):
    """The span-ordered folding DP (lanes own rows, whole dp triangle in LDS) in its two placements -- both foldings side
    by side in the node's workgroup (to ~165 columns each), or one folding per workgroup in a split launch (to 256
    columns) -- and, with DAFS_HIP_DD_SPAN=0, the column-owning forms it replaces on the same inputs; the dense case
    overflows the four candidate slots per column and takes the fallback.  All against the oracle, iteration log included."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.random_set(n, length, seed=77)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 77, density=density)
    want, (it, vi), got = _run_both(oracle, names, seqs, bp, t_max=t_max, slice_iters=7)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    monkeypatch.setenv("DAFS_HIP_DD_SPLIT", "0")  # span form only where both foldings fit the node's workgroup
    one = pipeline.run(names, seqs, bp=bp, t_max=t_max, skip_uncoupled_folds=False)
    assert one.output == want and one.dd_log == got.dd_log
    monkeypatch.setenv("DAFS_HIP_DD_SPAN", "0")   # the column-owning forms
    old = pipeline.run(names, seqs, bp=bp, t_max=t_max, level_sync=True, skip_uncoupled_folds=False)
    assert old.output == want and old.dd_log == got.dd_log


@pytest.mark.parametrize("length", [455, 600, 700])
def test_fast_folding_with_codes_in_hbm(oracle, monkeypatch, length):
    """Alignments of ~430-510 columns: the nibble table of the traceback no longer fits LDS beside the rows in
    flight, so the register form of the folding DP writes its codes to HBM (one byte per cell); from 513 to 768
    columns it runs on 48 lanes with up to 16 columns each.  Checked in the split placement (a workgroup per
    folding) and inside the leader's workgroup (one shared region, x then y)."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(4, length, seed=43)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    assert 420 < min(len(s) for s in seqs) and max(len(s) for s in seqs) <= 768
    bp = random_bp(seqs, 43, density=0.006)
    want, (it, vi), got = _run_both(oracle, names, seqs, bp, t_max=25)
    assert got.output == want
    monkeypatch.setenv("DAFS_HIP_DD_SPLIT", "0")
    ref = pipeline.run(names, seqs, bp=bp, t_max=25, level_sync=True, skip_uncoupled_folds=False)
    assert ref.output == want and ref.dd_log == got.dd_log


def test_uncoupled_nodes_without_their_folding_dps(oracle):
    """dafs_dd_params.skip_uncoupled_folds: a node without consensus base pairs keeps its alignment multipliers at
    zero whatever its two foldings do, so leaving the foldings out changes the iteration log of that node and
    nothing else -- the alignments, hence the output, are the reference's."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    for recs, seed in ((synth.random_set(10, 70, seed=51), 51), (synth.family_set(12, 60, seed=21), 21)):
        names, seqs = [r[0] for r in recs], [r[1] for r in recs]
        bp = random_bp(seqs, seed, density=0.03)
        want, (it, vi), full = _run_both(oracle, names, seqs, bp)
        lean = pipeline.run(names, seqs, bp=bp, skip_uncoupled_folds=True)
        assert full.output == want == lean.output
        for node, (its, viol, ncbp, score) in full.dd_log.items():
            if ncbp:  # coupled nodes are untouched
                assert lean.dd_log[node] == (its, viol, ncbp, score)
            else:
                assert lean.dd_log[node][0] <= its


def test_dense_base_pairs_overflow_the_register_form(oracle, monkeypatch):
    """With dense base-pair input a column collects more than DD_CAP candidate split points, the register form of the
    folding DP gives up and the result comes from the fallback: in a split launch the span-ordered form on the
    folder's whole workgroup (and no further register attempts in that launch), inside the leader's workgroup the
    wave form over HBM tables.  Both placements against the oracle."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(4, 260, seed=61)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 61, density=0.04)
    want, (it, vi), got = _run_both(oracle, names, seqs, bp, t_max=12, slice_iters=5)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    monkeypatch.setenv("DAFS_HIP_DD_SPLIT", "0")
    ref = pipeline.run(names, seqs, bp=bp, t_max=12, level_sync=True, skip_uncoupled_folds=False)
    assert ref.output == want and ref.dd_log == got.dd_log


def test_forced_iterations_equal_oracle(oracle):
    """force_iters (the bench mode that ignores the violated == 0 exit, dafs_dd_params / oracle pipeline.c:523): every
    node runs t_max iterations; output and iteration log must still be the oracle's."""
    from dafs_amd import pipeline
    from test_pct_gpu import random_bp
    recs = synth.family_set(7, 60, seed=31) + synth.random_set(2, 45, seed=32)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 31, density=0.04)
    for t_max in (1, 17):
        pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1, t_max=t_max, force_iters=1), bp=bp)
        pl.phase1(); pl.phase2()
        want = pl.output()
        it, vi = pl.dd_log()
        pl.close()
        got = pipeline.run(names, seqs, bp=bp, t_max=t_max, force_iters=1, skip_uncoupled_folds=False)
        assert got.output == want, t_max
        assert [int(x) for x in it] == [t_max] * (len(seqs) - 1)
        assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
        assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)


# ---- the dense decoder classes: Nussinov, NeedlemanWunsch ----
def test_dense_decoders_golden(ctx):
    """device rows straight against the reference-generated fixture (tests/golden/make_golden.py, oracle/_ref)"""
    z = np.load(os.path.join(G, "decoders_dense.npz"))
    for k in range(int(z["n"])):
        p, q, w, th = z["p%d" % k], z["q%d" % k], float(z["w%d" % k]), float(z["th%d" % k])
        s, ss = ctx.nussinov_dense(p, q, th, w)
        assert np.float32(s).tobytes() == np.float32(z["s%d" % k]).tobytes() and np.array_equal(ss, z["ss%d" % k]), k
        s, ss = ctx.nussinov_dense(p, None, th)
        assert np.float32(s).tobytes() == np.float32(z["sf%d" % k]).tobytes() and np.array_equal(ss, z["ssf%d" % k]), k
        pz, qz, tha = z["pz%d" % k], z["qz%d" % k], float(z["tha%d" % k])
        s, al = ctx.nw_dense(pz, qz, tha)
        assert np.float32(s).tobytes() == np.float32(z["sz%d" % k]).tobytes() and np.array_equal(al, z["al%d" % k]), k
        s, al = ctx.nw_dense(pz, None, tha)
        assert np.float32(s).tobytes() == np.float32(z["szf%d" % k]).tobytes() and np.array_equal(al, z["alf%d" % k]), k


def test_dense_decoders_fuzz_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(78)
    for t in range(20):
        L = int(rng.choice([1, 2, 3, 5, 31, 70, 130]))
        L2 = int(rng.choice([1, 2, 6, 40, 111]))
        dens = float(rng.choice([0.0, 0.05, 0.5]))
        p = (rng.random((L, L)) * (rng.random((L, L)) < dens)).astype(np.float32)
        if t % 2:
            p = (np.round(p * 4) / 4).astype(np.float32)
        q = ((rng.random((L, L)) - 0.4) * (rng.random((L, L)) < 0.3)).astype(np.float32)
        w, th = float(rng.choice([4.0, 1.3333334])), float(rng.choice([0.2, 0.05]))
        for qq in (q, None):
            a, b = ctx.nussinov_dense(p, qq, th, w), oracle.nussinov_dense(p, qq, th, w)
            assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), (t, L)
        pz = (rng.random((L, L2)) * (rng.random((L, L2)) < max(dens, 0.02))).astype(np.float32)
        qz = (rng.random((L, L2)) * (rng.random((L, L2)) < 0.2)).astype(np.float32)
        for qq in (qz, None):
            a, b = ctx.nw_dense(pz, qq, 0.01), oracle.nw_dense(pz, qq, 0.01)
            assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), (t, L, L2)


@pytest.mark.parametrize("level_sync", [False, True])
def test_lost_folders_are_recovered(level_sync):
    """Split mode (a node's two folding DPs on workgroups of their own) depends on all three workgroups being on the
    machine together.  DAFS_HIP_DD_LOSE_FOLDERS=1 makes every leader treat its folders as lost at its first collect:
    the node must be parked untouched, relaunched in the one-workgroup form and end with the same result -- not fail."""
    from dafs_amd import pipeline
    recs = synth.family_set(6, 300, seed=51)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    want = pipeline.run(names, seqs, level_sync=level_sync, skip_uncoupled_folds=False)
    assert max(max(d) for d in pipeline.run(names, seqs).dd_dims.values()) > 250   # wide enough for split mode
    os.environ["DAFS_HIP_DD_LOSE_FOLDERS"] = "1"
    try:
        got = pipeline.run(names, seqs, level_sync=level_sync, skip_uncoupled_folds=False)
    finally:
        os.environ.pop("DAFS_HIP_DD_LOSE_FOLDERS", None)
    assert got.output == want.output and got.dd_log == want.dd_log
    os.environ["DAFS_HIP_DD_SPLIT"] = "0"
    try:
        nosplit = pipeline.run(names, seqs, level_sync=level_sync, skip_uncoupled_folds=False)
    finally:
        os.environ.pop("DAFS_HIP_DD_SPLIT", None)
    assert nosplit.output == want.output and nosplit.dd_log == want.dd_log
