"""GPU parity: consistency transforms and sim_ (a14) against the oracle
pipeline's phase 1, bit for bit."""
import os

import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def random_bp(seqs, seed, density=0.03):
    rng = np.random.default_rng(seed)
    out = []
    for s in seqs:
        L = len(s)
        rows = []
        for i in range(L):
            js = [j for j in range(i + 1, L) if rng.random() < density]
            rows.append([(j, np.float32(0.011 + 0.9 * rng.random())) for j in js])
        rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint32)
        col = np.array([c for r in rows for c, _ in r], np.uint32)
        val = np.array([v for r in rows for _, v in r], np.float32)
        out.append((rp, col, val))
    return out


def check_phase1(oracle, names, seqs, bp, w_a=0.25, w_s=0.25, w_f=0.0):
    from dafs_amd import capi
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1, w_pct_a=w_a, w_pct_s=w_s, w_pct_f=w_f), bp=bp)
    pl.phase1()
    ctx = capi.Context(0)
    ctx.set_sequences(seqs)
    ctx.set_bp(bp)
    ctx.align_posteriors(fetch=False)
    if w_f != 0:
        ctx.fourway_consistency(w_f)  # replaces the un-relaxed store and the similarity scores (dafs.cpp:1808)
    assert ctx.sim().tobytes() == pl.sim().tobytes()
    ctx.consistency(w_a, w_s)
    n = len(seqs)
    mp = ctx.mp(1 if w_a != 0 else 0)
    p = 0
    for x in range(n):
        for y in range(x + 1, n):
            for tr, (a, b) in ((False, (x, y)), (True, (y, x))):
                rp, col, val = pl.mp(a, b)
                grp, gcol, gval = mp.csr(p, tr)
                assert np.array_equal(grp, rp), (x, y, tr)
                assert np.array_equal(gcol, col), (x, y, tr)
                assert gval.tobytes() == val.tobytes(), (x, y, tr, np.abs(gval - val).max())
            p += 1
    gbp = ctx.bp(1 if w_s != 0 else 0)
    for x in range(n):
        rp, col, val = pl.bp(x)
        assert np.array_equal(gbp[x][0], rp), x
        assert np.array_equal(gbp[x][1], col), x
        assert gbp[x][2].tobytes() == val.tobytes(), (x, np.abs(gbp[x][2] - val).max())
    ctx.close()
    pl.close()


def test_rf00005(oracle):
    import test_oracle_cpu as t
    recs = oracle.fasta(os.path.join(G, "RF00005_0.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    check_phase1(oracle, names, seqs, t._golden_bp(seqs))


@pytest.mark.parametrize("n,length,fam", [(6, 60, False), (9, 110, True), (3, 230, False), (2, 40, False)])
def test_synthetic(oracle, n, length, fam):
    recs = synth.family_set(n, length, seed=99) if fam else synth.random_set(n, length, seed=98)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    check_phase1(oracle, names, seqs, random_bp(seqs, 4))


def test_weights_off_and_negative(oracle):
    recs = synth.family_set(5, 50, seed=3)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 5)
    check_phase1(oracle, names, seqs, bp, w_a=0.0, w_s=0.25)
    check_phase1(oracle, names, seqs, bp, w_a=0.25, w_s=0.0)
    check_phase1(oracle, names, seqs, bp, w_a=-1.0, w_s=-1.0)


@pytest.mark.parametrize("w_f", [0.3, 1.0])
def test_fourway_consistency(oracle, w_f):
    """DAFS::relax_fourway_consistency (-f, dafs.cpp:377-444): the transformed matching rows, the similarity scores taken
    from them and both ordinary transforms on top, against the oracle pipeline (parity unpinned: the oracle restates
    dafs.cpp, which cannot be built here); and the whole run through the driver."""
    from dafs_amd import pipeline
    recs = synth.family_set(7, 80, seed=17)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    bp = random_bp(seqs, 6, density=0.05)
    check_phase1(oracle, names, seqs, bp, w_f=w_f)
    check_phase1(oracle, names, seqs, bp, w_a=0.0, w_s=0.0, w_f=w_f)  # the transformed rows themselves
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1, w_pct_f=w_f), bp=bp)
    pl.phase1(); pl.phase2()
    got = pipeline.run(names, seqs, bp=bp, w_pct_f=w_f, skip_uncoupled_folds=False)
    assert got.output == pl.output()
    pl.close()
