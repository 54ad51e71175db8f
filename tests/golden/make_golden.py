#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the REFERENCE's own code
(oracle/_ref/libdafs_ref.so, built by oracle/Makefile from /root/reference/src).
Run in the build container only (the reference does not exist on the GPU box):

    make -C oracle && python tests/golden/make_golden.py

Outputs are data only (inputs + expected outputs); no reference source is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from dafs_amd import synth  # noqa: E402


def csr_pack(items):
    """list of (rp, col, val) -> concatenated arrays + offsets"""
    rps, cols, vals, ro, eo = [], [], [], [0], [0]
    for rp, col, val in items:
        rps.append(rp); cols.append(col); vals.append(val)
        ro.append(ro[-1] + len(rp)); eo.append(eo[-1] + len(col))
    return dict(rowptr=np.concatenate(rps).astype(np.uint32), col=np.concatenate(cols).astype(np.uint32),
                val=np.concatenate(vals).astype(np.float32), rp_off=np.array(ro, np.int64), ent_off=np.array(eo, np.int64))


def main():
    ref = oracle_lib.load_ref()
    assert ref is not None, "build oracle/_ref first"
    rf5 = [s for _, s in ref.fasta(os.path.join(HERE, "RF00005_0.fa"))]
    syn = {L: [s for _, s in synth.random_set(3, L, seed=100 + L)] for L in (16, 80, 150)}
    odd = ["A", "CG", "acgu", "NNTTXXzA", "GGGAAACCCUUU"]  # no gap chars: CONTRAlign drops them (Sequence.cpp:87-92)

    pairs = [(a, b) for i, a in enumerate(rf5) for b in rf5[i + 1:]]
    for L in syn:
        s = syn[L]
        pairs += [(s[0], s[1]), (s[0], s[2]), (s[1], s[2])]
    pairs += [(a, b) for i, a in enumerate(odd) for b in odd[i + 1:]]

    for model, name in ((0, "probcons"), (1, "contralign")):
        th = 0.01
        packed = csr_pack([ref.align_calculate(a, b, th, model) for a, b in pairs])
        dense_fn = ref.probcons_posterior if model == 0 else ref.contralign_posterior
        dense = {"dense%d" % k: dense_fn(*pairs[k], 0.0) for k in (0, 7, 45, 48, 51)}
        np.savez_compressed(os.path.join(HERE, name + "_mp.npz"), th=np.float32(th),
                            seq1=np.array([a for a, _ in pairs]), seq2=np.array([b for _, b in pairs]),
                            dense_idx=np.array([0, 7, 45, 48, 51]), **packed, **dense)

    # CONTRAfold: full triangular posteriors
    fold_seqs = rf5 + [syn[80][0], syn[150][0], "GGGAAACCC", "ACGU", "A", "GGGGAAAACCCCNNTT"]
    post = [ref.contrafold_posterior(s) for s in fold_seqs]
    cons_seq = rf5[0]
    cons = "".join("?" if k % 7 else "." for k in range(len(cons_seq)))
    cons = "((" + cons[2:-2] + "))"
    np.savez_compressed(os.path.join(HERE, "contrafold_post.npz"), seqs=np.array(fold_seqs),
                        off=np.array([0] + list(np.cumsum([len(p) for p in post])), np.int64),
                        post=np.concatenate(post), cons_seq=np.array(cons_seq), cons_str=np.array(cons),
                        cons_post=ref.contrafold_posterior(cons_seq, cons))

    # decoders on random + real matrices
    rng = np.random.default_rng(2024)
    cases = {}
    k = 0
    for L, L2 in ((1, 1), (2, 3), (5, 4), (30, 33), (73, 74), (150, 141)):
        for dens in (0.0, 0.05, 0.3):
            p = (rng.random((L, L)) * (rng.random((L, L)) < dens)).astype(np.float32)
            q = ((rng.random((L, L)) - 0.3) * (rng.random((L, L)) < 0.3)).astype(np.float32)
            if k % 2:
                p = (np.round(p * 4) / 4).astype(np.float32)  # ties
            w, th = np.float32([4.0, 2.6666667, 1.0][k % 3]), np.float32([0.2, 0.01, 0.5][k % 3])
            s, ss = ref.nussinov(p, q, th, w)
            s2, ss2, br = ref.nussinov(p, None, th)
            pz = (rng.random((L, L2)) * (rng.random((L, L2)) < max(dens, 0.02))).astype(np.float32)
            qz = (rng.random((L, L2)) * (rng.random((L, L2)) < 0.3)).astype(np.float32)
            tha = np.float32([0.01, 0.2][k % 2])
            sz, al = ref.nw(pz, qz, tha)
            sz2, al2 = ref.nw(pz, None, tha)
            cases.update({"p%d" % k: p, "q%d" % k: q, "w%d" % k: w, "th%d" % k: th, "s%d" % k: s, "ss%d" % k: ss,
                          "sf%d" % k: s2, "ssf%d" % k: ss2, "br%d" % k: np.array(br),
                          "pz%d" % k: pz, "qz%d" % k: qz, "tha%d" % k: tha, "sz%d" % k: sz, "al%d" % k: al,
                          "szf%d" % k: sz2, "alf%d" % k: al2})
            k += 1
    np.savez_compressed(os.path.join(HERE, "decoders.npz"), n=np.int64(k), **cases)

    # the dense decoder classes (Nussinov, NeedlemanWunsch): same recipe, smaller sizes (the bifurcation loop is cubic)
    rng = np.random.default_rng(2025)
    cases = {}
    k = 0
    for L, L2 in ((1, 1), (2, 3), (4, 5), (17, 23), (60, 52), (96, 101)):
        for dens in (0.0, 0.08, 0.4):
            p = (rng.random((L, L)) * (rng.random((L, L)) < dens)).astype(np.float32)
            q = ((rng.random((L, L)) - 0.3) * (rng.random((L, L)) < 0.3)).astype(np.float32)
            if k % 2:
                p = (np.round(p * 4) / 4).astype(np.float32)  # ties
            w, th = np.float32([4.0, 2.6666667, 1.0][k % 3]), np.float32([0.2, 0.01, 0.5][k % 3])
            s, ss = ref.nussinov_dense(p, q, th, w)
            s2, ss2 = ref.nussinov_dense(p, None, th)
            pz = (rng.random((L, L2)) * (rng.random((L, L2)) < max(dens, 0.02))).astype(np.float32)
            qz = (rng.random((L, L2)) * (rng.random((L, L2)) < 0.3)).astype(np.float32)
            tha = np.float32([0.01, 0.2][k % 2])
            sz, al = ref.nw_dense(pz, qz, tha)
            sz2, al2 = ref.nw_dense(pz, None, tha)
            cases.update({"p%d" % k: p, "q%d" % k: q, "w%d" % k: w, "th%d" % k: th, "s%d" % k: s, "ss%d" % k: ss,
                          "sf%d" % k: s2, "ssf%d" % k: ss2, "pz%d" % k: pz, "qz%d" % k: qz, "tha%d" % k: tha,
                          "sz%d" % k: sz, "al%d" % k: al, "szf%d" % k: sz2, "alf%d" % k: al2})
            k += 1
    np.savez_compressed(os.path.join(HERE, "decoders_dense.npz"), n=np.int64(k), **cases)

    with open(os.path.join(HERE, "synth_checksums.txt"), "w") as f:
        for n, L, seed in ((32, 80, 12345), (128, 150, 12345), (256, 200, 12345), (512, 400, 12345)):
            f.write("random %d %d %d %s\n" % (n, L, seed, synth.checksum(synth.random_set(n, L, seed=seed, jitter=0.0 if L == 80 else 0.07))))
        for n, L, seed in ((32, 80, 12346), (128, 150, 12346)):
            f.write("family %d %d %d %s\n" % (n, L, seed, synth.checksum(synth.family_set(n, L, seed=seed))))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
