"""GPU parity: the HIP pair-HMM path (through the C ABI) against the oracle, bit for bit.
Covers rows a1-a7, a14, a15 of SURVEY.md section 8."""
import os

import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu

EX = os.path.join(os.path.dirname(__file__), "golden")


def _check_set(oracle, seqs, th=0.01, force_group=None, model=0):
    from dafs_amd import capi
    if force_group:
        os.environ["DAFS_HIP_FORCE_GROUP"] = str(force_group)
    else:
        os.environ.pop("DAFS_HIP_FORCE_GROUP", None)
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(seqs)
        res = ctx.align_posteriors(model, th)
    finally:
        ctx.close()
        os.environ.pop("DAFS_HIP_FORCE_GROUP", None)
    n = len(seqs)
    assert len(res) == n * (n - 1) // 2
    p = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert res.pair_x[p] == i and res.pair_y[p] == j
            rp, col, val = oracle.align_calculate(seqs[i], seqs[j], th, model)
            grp, gcol, gval = res.csr(p)
            assert np.array_equal(grp, rp), (i, j, "rowptr")
            assert np.array_equal(gcol, col), (i, j, "col")
            assert gval.tobytes() == val.tobytes(), (i, j, "val", np.abs(gval - val).max())
            # transposed rows: same entries, keyed by column, rows ascending (dafs.cpp:155-167)
            trp, tcol, tval = res.csr(p, transposed=True)
            L1, L2 = len(seqs[i]), len(seqs[j])
            rows = np.repeat(np.arange(L1, dtype=np.uint32), np.diff(rp))
            order = np.lexsort((rows, col))
            assert np.array_equal(tcol, rows[order]), (i, j, "tcol")
            assert tval.tobytes() == val[order].tobytes(), (i, j, "tval")
            assert np.array_equal(trp, np.concatenate([[0], np.cumsum(np.bincount(col, minlength=L2))]).astype(np.uint32))
            s = oracle.similarity(rp, col, val, L1, L2)
            assert np.float32(res.sim[p]).tobytes() == np.float32(s).tobytes(), (i, j, res.sim[p], s)
            p += 1


def test_rf00005_all_pairs(oracle):
    seqs = [s for _, s in oracle.fasta(os.path.join(EX, "RF00005_0.fa"))]
    _check_set(oracle, seqs)


@pytest.mark.parametrize("group", [16, 32, 64])
def test_every_group_width(oracle, group):
    seqs = [s for _, s in synth.random_set(7, 60, seed=7)]
    _check_set(oracle, seqs, force_group=group)


@pytest.mark.parametrize("n,length,seed", [(6, 150, 12345), (12, 80, 3), (4, 300, 5)])
def test_synthetic_sets(oracle, n, length, seed):
    seqs = [s for _, s in synth.random_set(n, length, seed=seed)]
    _check_set(oracle, seqs)


def test_family_set(oracle):
    seqs = [s for _, s in synth.family_set(8, 120, seed=12346)]
    _check_set(oracle, seqs)


def test_ragged_and_odd_residues(oracle):
    # lengths 1..5, mixed case, T/N and bytes outside the alphabet (class 'other')
    seqs = ["A", "CG", "acgu", "GGGAAACCC", "NNTTXX-zA", "ACGUACGUACGUACGUACGUACGUACGUACGUACGU", "u" * 17, "GCAUCGAUCGAUGCUAGCUAGCUAGCUAGCUAGCAUCGAUCAGCUAGCUAGCUAGCAUCAGCUAGCAUCAGCUAGUCGAUCAGCUAGC"]
    for g in (None, 16, 64):
        _check_set(oracle, seqs, force_group=g)
    _check_set(oracle, seqs, th=0.0)
    _check_set(oracle, seqs, th=0.25)


def test_pair_shard(oracle):
    from dafs_amd import capi
    seqs = [s for _, s in synth.random_set(9, 50, seed=11)]
    ctx = capi.Context(0)
    ctx.set_sequences(seqs)
    full = ctx.align_posteriors()
    part = ctx.align_posteriors(pair_begin=10, pair_end=23)
    ctx.close()
    assert len(part) == 13
    for k in range(13):
        assert part.pair_x[k] == full.pair_x[10 + k] and part.pair_y[k] == full.pair_y[10 + k]
        for tr in (False, True):
            a, b = part.csr(k, tr), full.csr(10 + k, tr)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert part.sim[k] == full.sim[10 + k]


# ---- CONTRAlign (rows a8, a9): same contract, 5-state model ----
def test_contralign_rf00005_all_pairs(oracle):
    seqs = [s for _, s in oracle.fasta(os.path.join(EX, "RF00005_0.fa"))]
    _check_set(oracle, seqs, model=1)


@pytest.mark.parametrize("group", [16, 32, 64])
def test_contralign_every_group(oracle, group):
    seqs = [s for _, s in synth.random_set(7, 60, seed=7)]
    _check_set(oracle, seqs, force_group=group, model=1)


@pytest.mark.parametrize("n,length,seed", [(6, 150, 12345), (4, 300, 5)])
def test_contralign_synthetic_sets(oracle, n, length, seed):
    seqs = [s for _, s in synth.random_set(n, length, seed=seed)]
    _check_set(oracle, seqs, model=1)


def test_contralign_ragged_and_odd_residues(oracle):
    seqs = ["A", "CG", "acgu", "GGGAAACCC", "NNTTXXzA", "ACGUACGUACGUACGUACGUACGUACGUACGUACGU", "u" * 17]
    for g in (None, 16):
        _check_set(oracle, seqs, force_group=g, model=1)
    _check_set(oracle, seqs, th=0.0, model=1)


def test_contralign_golden_rows():
    """the committed reference-generated rows, straight against the device"""
    from dafs_amd import capi
    z = np.load(os.path.join(EX, "contralign_mp.npz"))
    ctx = capi.Context(0)
    for k in (0, 7, 44, 45, 50, 54, 60):
        s1, s2 = str(z["seq1"][k]), str(z["seq2"][k])
        ctx.set_sequences([s1, s2])
        res = ctx.align_posteriors(capi.ALIGN_CONTRALIGN, float(z["th"]))
        rp, col, val = res.csr(0)
        r0, r1, e0, e1 = z["rp_off"][k], z["rp_off"][k + 1], z["ent_off"][k], z["ent_off"][k + 1]
        assert np.array_equal(rp, z["rowptr"][r0:r1]) and np.array_equal(col, z["col"][e0:e1])
        assert val.tobytes() == z["val"][e0:e1].tobytes()
    ctx.close()


def test_whole_run_contralign(oracle):
    import test_oracle_cpu as t
    from dafs_amd import capi, pipeline
    ka = t.known()
    recs = oracle.fasta(os.path.join(EX, "RF00005_0.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    got = pipeline.run(names, seqs, align_model=capi.ALIGN_CONTRALIGN)
    assert got.tree_line == ka["rf00005.contralign.tree"]
    assert len(got.rows[0]) == int(ka["rf00005.contralign.contrafold.columns"])
    assert got.rows[0] == ka["rf00005.contralign.contrafold.first_row"]
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=1))
    pl.phase1(); pl.phase2()
    assert got.output == pl.output()
    pl.close()


def test_set_mp_rebuilds_transposes_and_similarity():
    """dafs_hip_set_mp: from the forward rows alone the store equals the one the pair kernel wrote (transposed rows,
    similarity scores), so everything downstream is unchanged."""
    from dafs_amd import capi, synth
    recs = synth.family_set(9, 80, seed=3) + synth.random_set(3, 55, seed=4)
    seqs = [r[1] for r in recs]
    a = capi.Context(0)
    a.set_sequences(seqs)
    ref = a.align_posteriors()
    sim = a.sim()
    nnz, rps, cols, vals = [], [], [], []
    for p in range(len(ref)):
        rp, col, val = ref.csr(p)
        nnz.append(len(col)); rps.append(rp); cols.append(col); vals.append(val)
    b = capi.Context(0)
    b.set_sequences(seqs)
    b.set_mp(np.array(nnz, np.uint32), np.concatenate(rps), np.concatenate(cols), np.concatenate(vals))
    assert np.array_equal(b.sim().view(np.uint32), sim.view(np.uint32))
    got = b.mp(0)
    for p in range(len(ref)):
        for tr in (False, True):
            for u, v in zip(ref.csr(p, tr), got.csr(p, tr)):
                assert np.array_equal(np.asarray(u).view(np.uint32), np.asarray(v).view(np.uint32)), (p, tr)
    a.close(); b.close()
