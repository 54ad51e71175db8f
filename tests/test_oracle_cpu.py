"""CPU tests (no GPU): the oracle restatement against the golden vectors generated from the
reference's own code (tests/golden/make_golden.py) and against the known answers of the whole
pipeline (tests/golden/known_answers.txt).  Where oracle/_ref is present (build container) the
oracle is additionally fuzzed against it."""
import os

import numpy as np
import pytest

from dafs_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def known():
    d = {}
    for line in open(os.path.join(G, "known_answers.txt")):
        if line.startswith("#") or not line.strip():
            continue
        k, v = line.rstrip("\n").split("\t")
        d[k] = v
    return d


def _cases(npz):
    z = np.load(os.path.join(G, npz))
    for k in range(len(z["seq1"])):
        r0, r1 = z["rp_off"][k], z["rp_off"][k + 1]
        e0, e1 = z["ent_off"][k], z["ent_off"][k + 1]
        yield str(z["seq1"][k]), str(z["seq2"][k]), z["rowptr"][r0:r1], z["col"][e0:e1], z["val"][e0:e1]


def test_probcons_mp_golden(oracle):
    z = np.load(os.path.join(G, "probcons_mp.npz"))
    n = 0
    for s1, s2, rp, col, val in _cases("probcons_mp.npz"):
        orp, ocol, oval = oracle.align_calculate(s1, s2, float(z["th"]), 0)
        assert np.array_equal(orp, rp) and np.array_equal(ocol, col)
        assert oval.tobytes() == val.tobytes()
        n += 1
    assert n == 45 + 9 + 10
    pairs = list(zip(z["seq1"], z["seq2"]))
    for k in z["dense_idx"]:
        d = oracle.probcons_posterior(str(pairs[k][0]), str(pairs[k][1]), 0.0)
        assert d.tobytes() == z["dense%d" % k].tobytes()


def test_contralign_mp_golden(oracle):
    z = np.load(os.path.join(G, "contralign_mp.npz"))
    for s1, s2, rp, col, val in _cases("contralign_mp.npz"):
        orp, ocol, oval = oracle.align_calculate(s1, s2, float(z["th"]), 1)
        assert np.array_equal(orp, rp) and np.array_equal(ocol, col) and oval.tobytes() == val.tobytes()
    ka = known()
    seqs = [s for _, s in oracle.fasta(os.path.join(G, "RF00005_0.fa"))]
    rp, col, val = oracle.align_calculate(seqs[0], seqs[1], 0.01, 1)
    assert len(col) == int(ka["rf00005.seq0_seq1.contralign.nnz"])
    assert col[0] == 0 and "%.9g" % val[0] == ka["rf00005.seq0_seq1.contralign.first"]


def test_pipeline_rf00005_contralign_known_answers(oracle):
    """-a CONTRAlign -s CONTRAfold --no-alifold"""
    ka = known()
    recs = oracle.fasta(os.path.join(G, "RF00005_0.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=1))
    pl.phase1(); pl.phase2()
    lines = pl.output().split("\n")
    assert lines[0] == ka["rf00005.contralign.tree"]
    assert len(lines[4]) == int(ka["rf00005.contralign.contrafold.columns"])
    assert lines[4] == ka["rf00005.contralign.contrafold.first_row"]
    pl.close()


def test_pipeline_with_supplied_matching_rows(oracle):
    """orc_pipeline_set_mp (the --align-aux path of the checker, align_model = 2): rows handed in equal rows computed"""
    recs = oracle.fasta(os.path.join(G, "RF00005_0.fa"))[:6]
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    for model in (0, 1):
        a = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=model))
        a.phase1(); a.phase2()
        b = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=2),
                            mp=lambda x, y: oracle.align_calculate(seqs[x], seqs[y], 0.01, model))
        b.phase1(); b.phase2()
        assert a.output() == b.output() and a.output().count("\n") == 3 + 2 * len(seqs)
        assert [list(v) for v in a.dd_log()] == [list(v) for v in b.dd_log()]
        a.close(); b.close()
    c = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=2))   # rows missing: refused, not computed
    assert oracle.lib.orc_pipeline_phase1(c.h) != 0
    c.close()


def test_parallel_oracle_run_equals_the_serial_one(oracle):
    """oracle_lib.parallel_oracle_run (models in worker processes, fed through the --fold-aux / --align-aux paths) is the
    serial pipeline, text and iteration log"""
    import oracle_lib
    recs = synth.family_set(9, 50, seed=3) + synth.random_set(3, 40, seed=4)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    for model in (0, 1):
        out, (it, vi) = oracle_lib.parallel_oracle_run(names, seqs, workers=3, align_model=model)
        pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0, align_model=model))
        pl.phase1(); pl.phase2()
        assert out == pl.output()
        sit, svi = pl.dd_log()
        assert list(it) == list(sit) and list(vi) == list(svi)
        pl.close()


def test_probcons_known_scalars(oracle):
    ka = known()
    seqs = [s for _, s in oracle.fasta(os.path.join(G, "RF00005_0.fa"))]
    rp, col, val = oracle.align_calculate(seqs[0], seqs[1], 0.01, 0)
    assert len(col) == int(ka["rf00005.seq0_seq1.probcons.nnz"])
    assert col[0] == 0 and "%.9g" % val[0] == ka["rf00005.seq0_seq1.probcons.first"]


def test_decoders_golden(oracle):
    z = np.load(os.path.join(G, "decoders.npz"))
    for k in range(int(z["n"])):
        p, q, w, th = z["p%d" % k], z["q%d" % k], float(z["w%d" % k]), float(z["th%d" % k])
        s, ss = oracle.nussinov(p, q, th, w)
        assert np.float32(s).tobytes() == np.float32(z["s%d" % k]).tobytes() and np.array_equal(ss, z["ss%d" % k])
        s, ss = oracle.nussinov(p, None, th)
        assert np.float32(s).tobytes() == np.float32(z["sf%d" % k]).tobytes() and np.array_equal(ss, z["ssf%d" % k])
        import ctypes as C
        buf = C.create_string_buffer(len(ss) + 1)
        oracle.lib.orc_make_brackets(len(ss), ss.ctypes.data, buf)
        assert buf.value.decode() == str(z["br%d" % k])
        pz, qz, tha = z["pz%d" % k], z["qz%d" % k], float(z["tha%d" % k])
        s, al = oracle.nw(pz, qz, tha)
        assert np.float32(s).tobytes() == np.float32(z["sz%d" % k]).tobytes() and np.array_equal(al, z["al%d" % k])
        s, al = oracle.nw(pz, None, tha)
        assert np.float32(s).tobytes() == np.float32(z["szf%d" % k]).tobytes() and np.array_equal(al, z["alf%d" % k])


def test_dense_decoders_golden(oracle):
    """the restatement of the dense classes (Nussinov, NeedlemanWunsch) against the reference-generated fixture"""
    z = np.load(os.path.join(G, "decoders_dense.npz"))
    for k in range(int(z["n"])):
        p, q, w, th = z["p%d" % k], z["q%d" % k], float(z["w%d" % k]), float(z["th%d" % k])
        s, ss = oracle.nussinov_dense(p, q, th, w)
        assert np.float32(s).tobytes() == np.float32(z["s%d" % k]).tobytes() and np.array_equal(ss, z["ss%d" % k]), k
        s, ss = oracle.nussinov_dense(p, None, th)
        assert np.float32(s).tobytes() == np.float32(z["sf%d" % k]).tobytes() and np.array_equal(ss, z["ssf%d" % k]), k
        pz, qz, tha = z["pz%d" % k], z["qz%d" % k], float(z["tha%d" % k])
        s, al = oracle.nw_dense(pz, qz, tha)
        assert np.float32(s).tobytes() == np.float32(z["sz%d" % k]).tobytes() and np.array_equal(al, z["al%d" % k]), k
        s, al = oracle.nw_dense(pz, None, tha)
        assert np.float32(s).tobytes() == np.float32(z["szf%d" % k]).tobytes() and np.array_equal(al, z["alf%d" % k]), k


def test_dense_decoders_fuzz_vs_reference(oracle, ref):
    rng = np.random.default_rng(5)
    for t in range(12):
        L, L2 = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        p = (rng.random((L, L)) * (rng.random((L, L)) < 0.2)).astype(np.float32)
        q = ((rng.random((L, L)) - 0.4) * (rng.random((L, L)) < 0.3)).astype(np.float32)
        for qq in (q, None):
            a, b = oracle.nussinov_dense(p, qq, 0.1, 3.0), ref.nussinov_dense(p, qq, 0.1, 3.0)
            assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), t
        pz = (rng.random((L, L2)) * (rng.random((L, L2)) < 0.1)).astype(np.float32)
        qz = (rng.random((L, L2)) * (rng.random((L, L2)) < 0.2)).astype(np.float32)
        for qq in (qz, None):
            a, b = oracle.nw_dense(pz, qq, 0.01), ref.nw_dense(pz, qq, 0.01)
            assert np.float32(a[0]).tobytes() == np.float32(b[0]).tobytes() and np.array_equal(a[1], b[1]), t


def _orc_fold_post(oracle, s, cons=None):
    L = len(s)
    out = np.zeros((L + 1) * (L + 2) // 2, np.float32)
    rc = oracle.lib.orc_contrafold_posterior(s.encode(), L, None if cons is None else cons.encode(), out.ctypes.data)
    assert rc == len(out)
    return out


def test_contrafold_golden(oracle):
    z = np.load(os.path.join(G, "contrafold_post.npz"))
    for k, s in enumerate(z["seqs"]):
        got = _orc_fold_post(oracle, str(s))
        assert got.tobytes() == z["post"][z["off"][k]:z["off"][k + 1]].tobytes(), (k, len(str(s)))
    got = _orc_fold_post(oracle, str(z["cons_seq"]), str(z["cons_str"]))
    assert got.tobytes() == z["cons_post"].tobytes()
    ka = known()
    p0 = z["post"][z["off"][0]:z["off"][1]]
    assert (p0 > np.float32(0.01)).sum() == int(ka["rf00005.seq0.contrafold.nnz_gt_0.01"])
    assert "%.9g" % p0.max() == ka["rf00005.seq0.contrafold.max"]


def test_fold_adapter_matches_golden_rows(oracle):
    """orc_fold_calculate (fold.cpp:174-189) == rows derived from the golden triangular posteriors"""
    seqs = [s for _, s in oracle.fasta(os.path.join(G, "RF00005_0.fa"))][:4]
    for s, (rp, col, val) in zip(seqs, _golden_bp(seqs)):
        L = len(s)
        orp = np.zeros(L + 1, np.uint32); ocol = np.zeros(L * L, np.uint32); oval = np.zeros(L * L, np.float32)
        n = oracle.lib.orc_fold_calculate(s.encode(), L, None, 0.01, orp.ctypes.data, ocol.ctypes.data, oval.ctypes.data)
        assert n == len(col) and np.array_equal(orp, rp) and np.array_equal(ocol[:n], col) and oval[:n].tobytes() == val.tobytes()


def _golden_bp(seqs):
    """CONTRAfold BP rows (p > 0.01, fold.cpp:181-188) from the golden triangular posteriors."""
    z = np.load(os.path.join(G, "contrafold_post.npz"))
    zs = [str(s) for s in z["seqs"]]
    out = []
    for s in seqs:
        k = zs.index(s)
        post = z["post"][z["off"][k]:z["off"][k + 1]]
        L = len(s)
        rows = [[] for _ in range(L)]
        t = 0
        for i in range(L + 1):
            for j in range(i, L + 1):
                if i != 0 and post[t] > np.float32(0.01):
                    rows[i - 1].append((j - 1, post[t]))
                t += 1
        rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint32)
        col = np.array([c for r in rows for c, _ in r], np.uint32)
        val = np.array([v for r in rows for _, v in r], np.float32)
        out.append((rp, col, val))
    return out


def test_pipeline_rf00005_known_answers(oracle):
    """README.md:59 tree + SURVEY Appendix C rows (-s CONTRAfold --no-alifold)."""
    ka = known()
    recs = oracle.fasta(os.path.join(G, "RF00005_0.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0))  # CONTRAfold computed by the oracle itself
    pl.phase1()
    assert pl.output().split("\n")[0] == ka["rf00005.probcons.tree"]
    for x, (rp, col, val) in enumerate(_golden_bp(seqs)):
        pass  # (the un-relaxed rows are checked in test_fold_adapter_matches_golden_rows)
    pl.phase2()
    lines = pl.output().split("\n")
    assert lines[1] == ">SS_cons"
    rows = lines[3:]
    assert len(rows[1]) == int(ka["rf00005.probcons.contrafold.columns"])
    assert rows[0] == "> " + names[0]
    assert rows[1] == ka["rf00005.probcons.contrafold.first_row"]
    assert rows[2 * 9 + 1] == ka["rf00005.probcons.contrafold.last_row"]
    it, vi = pl.dd_log()
    assert len(it) == 9 and (vi == 0).all()
    pl.close()


def test_pipeline_rf00017_tree(oracle):
    ka = known()
    recs = oracle.fasta(os.path.join(G, "RF00017_4.fa"))
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    empty = [(np.zeros(len(s) + 1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32)) for s in seqs]
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=1), bp=empty)
    pl.phase1()
    assert pl.output().split("\n")[0] == ka["rf00017.probcons.tree"]
    pl.close()


def test_fasta_reader(oracle, tmp_path):
    f = tmp_path / "x.fa"
    f.write_text(">a b c\nACGU\nacgu123\n(((...)))\n>second\nGG-CC\n\n>third\nNNNN\n")
    recs = oracle.fasta(str(f))
    assert recs == [("a b c", "ACGUacgu"), ("second", "GG"), ("third", "NNNN")]


def test_synth_checksums():
    for line in open(os.path.join(G, "synth_checksums.txt")):
        kind, n, L, seed, h = line.split()
        n, L, seed = int(n), int(L), int(seed)
        if kind == "random":
            recs = synth.random_set(n, L, seed=seed, jitter=0.0 if L == 80 else 0.07)
        else:
            recs = synth.family_set(n, L, seed=seed)
        assert synth.checksum(recs) == h
        assert len(recs) == n


def test_oracle_vs_reference_fuzz(oracle, ref):
    rng = np.random.default_rng(5)
    for t in range(40):
        L1, L2 = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        alpha = "ACGU" if t % 3 else "ACGUTNacgux-"
        s1 = "".join(rng.choice(list(alpha), L1)); s2 = "".join(rng.choice(list(alpha), L2))
        for th in (0.0, 0.01):
            assert oracle.probcons_posterior(s1, s2, th).tobytes() == ref.probcons_posterior(s1, s2, th).tobytes()
        if t < 12:
            assert _orc_fold_post(oracle, s1).tobytes() == ref.contrafold_posterior(s1).tobytes()
        a, b = oracle.align_calculate(s1, s2, 0.01, 0), ref.align_calculate(s1, s2, 0.01, 0)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    assert oracle.fasta(os.path.join(G, "RF00017_4.fa")) == ref.fasta(os.path.join(G, "RF00017_4.fa"))
