"""The headline workload at its full size (BASELINE.json: N=128, L~150; 8128 pairs) on the GPU.
The oracle is too slow to redo all of it inside a test, so the checks are (a) a strided sample of pairs
against the oracle, bit for bit, and (b) properties that hold for every pair regardless of size:
forward and transposed rows hold the same entries, columns ascend, values lie in (th, 1], the similarity
matrix is symmetric with a unit diagonal -- and, for the whole run, that the aligned rows spell the input
sequences, all rows have the consensus length, the structure is a proper pairing, and the resident-node
schedule reproduces the level-synchronous one."""
import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu
N, L = 128, 150


@pytest.fixture(scope="module")
def headline():
    recs = synth.random_set(N, L, seed=12345)
    return [r[0] for r in recs], [r[1] for r in recs]


def test_all_pairs_properties_and_sample(oracle, headline):
    from dafs_amd import capi
    names, seqs = headline
    ctx = capi.Context(0)
    try:
        ctx.set_sequences(seqs)
        res = ctx.align_posteriors(0, 0.01)
        sim = ctx.sim()
    finally:
        ctx.close()
    npairs = N * (N - 1) // 2
    assert len(res) == npairs
    assert np.array_equal(sim, sim.T) and np.all(np.diag(sim) == 1) and np.all(sim > 0) and np.all(sim <= 1)
    checked = 0
    for p in range(npairs):
        x, y = int(res.pair_x[p]), int(res.pair_y[p])
        l1, l2 = len(seqs[x]), len(seqs[y])
        rp, col, val = res.csr(p)
        trp, tcol, tval = res.csr(p, transposed=True)
        assert rp[0] == 0 and rp[-1] == len(col) == len(tcol) and len(rp) == l1 + 1 and len(trp) == l2 + 1
        assert np.all(val > np.float32(0.01)) and np.all(val <= 1)
        rows = np.repeat(np.arange(l1, dtype=np.uint32), np.diff(rp))
        same_row = rows[1:] == rows[:-1]
        assert np.all(col[1:][same_row] > col[:-1][same_row])          # columns ascend within a row
        order = np.lexsort((rows, col))
        assert np.array_equal(tcol, rows[order]) and tval.tobytes() == val[order].tobytes()
        if p % 61 == 0:                                                  # the oracle on a strided sample
            orp, ocol, oval = oracle.align_calculate(seqs[x], seqs[y], 0.01, 0)
            assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and val.tobytes() == oval.tobytes(), (x, y)
            assert np.float32(res.sim[p]).tobytes() == np.float32(oracle.similarity(orp, ocol, oval, l1, l2)).tobytes()
            checked += 1
    assert checked >= 130


def test_whole_run_properties(headline):
    from dafs_amd import pipeline
    names, seqs = headline
    a = pipeline.run(names, seqs)
    b = pipeline.run(names, seqs, level_sync=True)
    assert a.output == b.output and a.dd_log == b.dd_log           # scheduling does not change a bit
    width = len(a.ss_str)
    assert len(a.rows) == N and all(len(r) == width for r in a.rows)
    assert sorted(r.replace("-", "") for r in a.rows) == sorted(seqs)
    assert not any(all(r[c] == "-" for r in a.rows) for c in range(width))   # no all-gap column
    depth = 0
    for ch in a.ss_str:
        assert ch in "().", ch
        depth += ch == "("
        depth -= ch == ")"
        assert depth >= 0
    assert depth == 0
    its = [v[0] for v in a.dd_log.values()]
    assert len(its) == N - 1 and max(its) <= 600


def test_whole_run_equals_oracle(headline):
    """BASELINE config 3 end to end against the CPU port, bit for bit: tree line, structure line, every alignment row and the
    per-node iteration log (parity unpinned for the dafs.cpp half).  The oracle's models run process-parallel
    (oracle_lib.parallel_oracle_run), its transforms, tree and progressive phase on one core."""
    import oracle_lib
    from dafs_amd import pipeline
    names, seqs = headline
    want, (it, vi) = oracle_lib.parallel_oracle_run(names, seqs)
    got = pipeline.run(names, seqs, skip_uncoupled_folds=False)
    assert got.output == want
    assert sorted(v[0] for v in got.dd_log.values()) == sorted(int(x) for x in it)
    assert sorted(v[1] for v in got.dd_log.values()) == sorted(int(x) for x in vi)
    assert pipeline.run(names, seqs).output == want          # the drivers' default (uncoupled foldings left out)
