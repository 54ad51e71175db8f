"""GPU parity: the HIP pair-HMM path (through the C ABI) against the oracle, bit for bit.
Covers rows a1-a7, a14, a15 of SURVEY.md section 8."""
import os

import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu

EX = os.path.join(os.path.dirname(__file__), "golden")


def _check_set(oracle, seqs, th=0.01, force_group=None):
    from dafs_amd import capi
    if force_group:
        os.environ["DAFS_HIP_FORCE_GROUP"] = str(force_group)
    else:
        os.environ.pop("DAFS_HIP_FORCE_GROUP", None)
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(seqs)
        res = ctx.align_posteriors(capi.ALIGN_PROBCONS, th)
    finally:
        ctx.close()
        os.environ.pop("DAFS_HIP_FORCE_GROUP", None)
    n = len(seqs)
    assert len(res) == n * (n - 1) // 2
    p = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert res.pair_x[p] == i and res.pair_y[p] == j
            rp, col, val = oracle.align_calculate(seqs[i], seqs[j], th)
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
