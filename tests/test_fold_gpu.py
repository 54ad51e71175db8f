"""GPU parity: CONTRAfold inside/outside/posterior kernel against the golden
vectors generated from the reference and against the oracle, bit for bit."""
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


def test_golden_posteriors(ctx):
    z = np.load(os.path.join(G, "contrafold_post.npz"))
    for k, s in enumerate(z["seqs"]):
        post, logz = ctx.fold_posterior_dense(str(s))
        want = z["post"][z["off"][k]:z["off"][k + 1]]
        assert post.tobytes() == want.tobytes(), (k, len(str(s)), np.abs(post - want).max(), int((post != want).sum()))


def test_golden_constrained(ctx):
    z = np.load(os.path.join(G, "contrafold_post.npz"))
    post, _ = ctx.fold_posterior_dense(str(z["cons_seq"]), str(z["cons_str"]))
    assert post.tobytes() == z["cons_post"].tobytes()


def test_fuzz_vs_oracle(ctx, oracle):
    import test_oracle_cpu as t
    rng = np.random.default_rng(11)
    for k in range(14):
        L = int(rng.choice([1, 2, 3, 5, 9, 31, 33, 64, 100, 181]))
        s = "".join(rng.choice(list("ACGU" if k % 3 else "ACGUTNacgu-"), L))
        post, _ = ctx.fold_posterior_dense(s)
        assert post.tobytes() == t._orc_fold_post(oracle, s).tobytes(), (k, L)
    # constraints: unpaired stretches and forced pairs
    s = "GGGAAACUUCGGUUUCCCAAGGGAAACCC"
    for cons in ("?" * len(s), "." * 3 + "?" * (len(s) - 3), "(" + "?" * (len(s) - 2) + ")", "((?..........?))" + "?" * (len(s) - 16)):
        post, _ = ctx.fold_posterior_dense(s, cons)
        assert post.tobytes() == t._orc_fold_post(oracle, s, cons).tobytes(), cons


def test_sequences_too_long_for_the_lds_ring(ctx, oracle, monkeypatch):
    """Beyond ~540 nt the 33 most recent spans of FC/FCo no longer fit LDS and the single-branch terms read them
    from HBM/L2 instead.  One sequence of that length against the oracle, and the short fuzz set with that form
    forced (DAFS_HIP_CF_NORING)."""
    import test_oracle_cpu as t
    rng = np.random.default_rng(5)
    s = "".join(rng.choice(list("ACGU"), 600))
    post, _ = ctx.fold_posterior_dense(s)
    assert post.tobytes() == t._orc_fold_post(oracle, s).tobytes()
    monkeypatch.setenv("DAFS_HIP_CF_NORING", "1")
    for k, L in enumerate([3, 9, 33, 64, 100, 181]):
        s = "".join(rng.choice(list("ACGU" if k % 2 else "ACGUTN"), L))
        post, _ = ctx.fold_posterior_dense(s)
        assert post.tobytes() == t._orc_fold_post(oracle, s).tobytes(), L
    s = "GGGAAACUUCGGUUUCCCAAGGGAAACCC"
    cons = "((?..........?))" + "?" * (len(s) - 16)
    post, _ = ctx.fold_posterior_dense(s, cons)
    assert post.tobytes() == t._orc_fold_post(oracle, s, cons).tobytes()


def test_term_pool_forms(ctx, oracle, monkeypatch):
    """The single-branch terms of a span are evaluated into an LDS pool before the cells fold them (k_contrafold).  Every
    form must give the oracle's bits: lists that overflow the pool (a GC repeat pairs everywhere: ~250 terms for every
    second cell, several times what the pool holds, so most cells walk their partners themselves), the pool without
    the FC ring beside it (from ~330 nt on), one buffer instead of two, no pool at all, and other splits of the
    wavefronts between cells and terms."""
    import test_oracle_cpu as t
    rng = np.random.default_rng(17)
    dense = ["GC" * 90, "GU" * 70 + "ACGU" * 10, "G" * 60 + "C" * 60]
    for s in dense:
        post, _ = ctx.fold_posterior_dense(s)
        assert post.tobytes() == t._orc_fold_post(oracle, s).tobytes(), s[:8]
    s400 = "".join(rng.choice(list("ACGU"), 400))
    post, _ = ctx.fold_posterior_dense(s400)
    want400 = t._orc_fold_post(oracle, s400)
    assert post.tobytes() == want400.tobytes()
    fuzz = ["".join(rng.choice(list("ACGU" if k % 2 else "ACGUTN"), L)) for k, L in enumerate([2, 3, 9, 33, 64, 100, 181])] + [dense[0]]
    want = [t._orc_fold_post(oracle, s) for s in fuzz]
    for env in ({"DAFS_HIP_CF_NOPOOL": "1"}, {"DAFS_HIP_CF_NOOVERLAP": "1"}, {"DAFS_HIP_CF_CELL_WAVES": "16"},
                {"DAFS_HIP_CF_CELL_WAVES": "3"}, {"DAFS_HIP_CF_THREADS": "256"}, {"DAFS_HIP_CF_NORING": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for s, w in zip(fuzz, want):
            post, _ = ctx.fold_posterior_dense(s)
            assert post.tobytes() == w.tobytes(), (env, len(s))
        for k in env:
            monkeypatch.delenv(k)


def test_batch_rows_vs_oracle(ctx, oracle):
    seqs = [s for _, s in synth.random_set(5, 70, seed=21)] + ["ACGU", "GGGAAACCCTTNN"]
    ctx.set_sequences(seqs)
    ctx.fold_posteriors(0.01)
    got = ctx.bp(0)
    for s, (rp, col, val) in zip(seqs, got):
        L = len(s)
        orp = np.zeros(L + 1, np.uint32); ocol = np.zeros(L * L + 1, np.uint32); oval = np.zeros(L * L + 1, np.float32)
        n = oracle.lib.orc_fold_calculate(s.encode(), L, None, 0.01, orp.ctypes.data, ocol.ctypes.data, oval.ctypes.data)
        assert np.array_equal(rp, orp) and np.array_equal(col, ocol[:n]) and val.tobytes() == oval[:n].tobytes()


def test_whole_run_with_device_fold(oracle):
    """dafs -s CONTRAfold --no-alifold on RF00005: tree == README.md:59, rows == SURVEY Appendix C"""
    import test_oracle_cpu as t
    from dafs_amd import pipeline
    ka = t.known()
    recs = oracle.fasta(os.path.join(G, "RF00005_0.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    got = pipeline.run(names, seqs, skip_uncoupled_folds=False)
    assert got.tree_line == ka["rf00005.probcons.tree"]
    assert len(got.rows[0]) == int(ka["rf00005.probcons.contrafold.columns"])
    assert got.rows[0] == ka["rf00005.probcons.contrafold.first_row"]
    assert got.rows[-1] == ka["rf00005.probcons.contrafold.last_row"]
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0))
    pl.phase1(); pl.phase2()
    assert got.output == pl.output()
    pl.close()
