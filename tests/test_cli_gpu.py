"""GPU tests of the C++ host side: the `dafs` command line (drop-in for the reference's CLI and
output format) and the plugin classes that implement the reference's
four plugin interfaces over the C ABI (row b)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from dafs_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
DAFS = os.path.join(ROOT, "dafs_amd", "dafs")
SELFTEST = os.path.join(ROOT, "dafs_amd", "plugin_selftest")


def run_cli(*args):
    r = subprocess.run([DAFS] + list(args), capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout, r.stderr


def oracle_output(oracle, path, **kw):
    recs = oracle.fasta(path)
    names, seqs = [n for n, _ in recs], [s for _, s in recs]
    pl = oracle.pipeline(names, seqs, oracle.params(fold_model=0, **kw))
    pl.phase1(); pl.phase2()
    out = pl.output()
    it, _ = pl.dd_log()
    pl.close()
    return out, it


def test_rf00005_default_and_known_answers(oracle):
    import test_oracle_cpu as t
    ka = t.known()
    rc, out, err = run_cli("-s", "CONTRAfold", "--no-alifold", os.path.join(G, "RF00005_0.fa"))
    assert rc == 0, err
    lines = out.split("\n")
    assert lines[0] == ka["rf00005.probcons.tree"]          # README.md:59
    assert lines[4] == ka["rf00005.probcons.contrafold.first_row"]
    assert lines[22] == ka["rf00005.probcons.contrafold.last_row"]
    want, _ = oracle_output(oracle, os.path.join(G, "RF00005_0.fa"))
    assert out == want


def test_contralign_and_flags(oracle):
    path = os.path.join(G, "RF00005_0.fa")
    rc, out, err = run_cli("-a", "CONTRAlign", "--no-alifold", path)
    assert rc == 0, err
    assert out == oracle_output(oracle, path, align_model=1)[0]
    rc, out, err = run_cli("-w", "2.0", "--eta=0.25", "-m", "50", "-p0.1", "-q", "0.4", "-u", "0.02", "-t", "0.3", path)
    assert rc == 0, err
    want, _ = oracle_output(oracle, path, w=2.0, eta0=0.25, t_max=50, w_pct_a=0.1, w_pct_s=0.4, th_a=0.02, th_s=0.3, th_s1=0.3)
    assert out == want
    rc, out, err = run_cli("-g", "4", "-G", "1", path)  # thresholds 1/(1+gamma) = 0.2 and 0.5
    assert rc == 0, err
    assert out == oracle_output(oracle, path, th_s=np.float32(1.0 / 5.0), th_s1=np.float32(0.5))[0]


def test_fourway_and_bp_update_flags(oracle, tmp_path):
    """-f (relax_fourway_consistency, dafs.cpp:377-444), --bp-update (root node, :919-934) and --bp-update1 (final structure,
    :1863-1869) through the command line and through the Python driver, against the oracle pipeline.  Parity unpinned:
    the oracle restates dafs.cpp, which cannot be built in this image."""
    from dafs_amd import pipeline
    recs = synth.family_set(6, 70, seed=31)
    fa = tmp_path / "fam.fa"
    fa.write_text(synth.to_fasta(recs))
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    for args, kw, pkw in ((["-f", "0.4"], dict(w_pct_f=0.4), dict(w_pct_f=0.4)),
                          (["--bp-update"], dict(bp_update=1), dict(bp_update=True)),
                          (["--bp-update1"], dict(bp_update1=1), dict(bp_update1=True)),
                          (["-f", "0.2", "--bp-update", "--bp-update1", "-m", "40"], dict(w_pct_f=0.2, bp_update=1, bp_update1=1, t_max=40),
                           dict(w_pct_f=0.2, bp_update=True, bp_update1=True, t_max=40))):
        want, _ = oracle_output(oracle, str(fa), **kw)
        rc, out, err = run_cli(*(args + [str(fa)]))
        assert rc == 0, err
        assert out == want, args
        assert pipeline.run(names, seqs, **pkw).output == want, args


def test_synthetic_family_verbose_log(oracle, tmp_path):
    recs = synth.family_set(12, 90, seed=5)
    fa = tmp_path / "fam.fa"
    fa.write_text(synth.to_fasta(recs))
    rc, out, err = run_cli("-v", "1", str(fa))
    assert rc == 0, err
    want, iters = oracle_output(oracle, str(fa))
    assert out == want
    logged = sorted(int(l.split(",")[0].split()[1]) for l in err.splitlines() if l.startswith("Step:"))
    assert logged == sorted(int(x) for x in iters)          # dafs.cpp:1292 per-node log


def test_fold_aux_round_trip(tmp_path):
    path = os.path.join(G, "RF00005_0.fa")
    aux = tmp_path / "bp.aux"
    rc, out1, err = run_cli("--save-fold-aux", str(aux), "--save-align-aux", str(tmp_path / "mp.aux"), path)
    assert rc == 0, err
    rc, out2, err = run_cli("--fold-aux", str(aux), path)
    assert rc == 0, err
    assert out1 == out2                                      # 9 significant digits round-trip float32 exactly
    first = (tmp_path / "mp.aux").read_text().splitlines()[:2]
    assert first[0] == "> 1 2" and first[1].startswith("1 1:0.661191404")


def test_align_aux_round_trip(tmp_path):
    """--align-aux (AUXAlign, src/align.cpp:204-246): supplied rows reproduce the run that exported them -- the
    transposes and the similarity scores are rebuilt from the rows alone (dafs_hip_set_mp)."""
    recs = synth.family_set(7, 70, seed=15)
    fa = tmp_path / "f.fa"
    fa.write_text(synth.to_fasta(recs))
    aux = tmp_path / "mp.aux"
    rc, out1, err = run_cli("--save-align-aux", str(aux), str(fa))
    assert rc == 0, err
    rc, out2, err = run_cli("--align-aux", str(aux), str(fa))
    assert rc == 0, err
    assert out1 == out2
    rc, out3, err = run_cli("-a", "CONTRAlign", "--align-aux", str(aux), str(fa))  # the model is not consulted
    assert rc == 0 and out3 == out1, err


def test_align_aux_file_read_by_the_reference_reader(ref, tmp_path):
    """The wire format pinned from the other side: the file `dafs --save-align-aux` writes is parsed by the reference's
    own AUXAlign::calculate (src/align.cpp:204-246, compiled into oracle/_ref), and what it loads equals the rows the
    device computed, bit for bit (9 significant digits carry a float32 exactly)."""
    from dafs_amd import capi
    recs = synth.family_set(6, 55, seed=25) + synth.random_set(2, 40, seed=26)
    seqs = [r[1] for r in recs]
    fa = tmp_path / "f.fa"
    fa.write_text(synth.to_fasta(recs))
    for model, flag in ((capi.ALIGN_PROBCONS, "ProbCons"), (capi.ALIGN_CONTRALIGN, "CONTRAlign")):
        aux = tmp_path / ("mp_%s.aux" % flag)
        rc, out, err = run_cli("-a", flag, "--save-align-aux", str(aux), str(fa))
        assert rc == 0, err
        nnz, rowptr, col, val = ref.auxalign_load(str(aux), seqs)
        ctx = capi.Context(0)
        ctx.set_sequences(seqs)
        dev = ctx.align_posteriors(model, 0.01)
        ctx.close()
        r0 = e0 = 0
        for p in range(len(dev)):
            rp, c, v = dev.csr(p)
            l1 = len(seqs[int(dev.pair_x[p])])
            assert nnz[p] == len(c), p
            assert np.array_equal(rowptr[r0:r0 + l1 + 1], rp), p
            assert np.array_equal(col[e0:e0 + len(c)], c) and val[e0:e0 + len(c)].tobytes() == v.tobytes(), p
            r0 += l1 + 1
            e0 += len(c)
        assert e0 == len(col)


def test_single_sequence_and_refinement(tmp_path):
    one = tmp_path / "one.fa"
    one.write_text(">only\nGGGAAACCCUUUAGGGCCC\n")
    rc, out, err = run_cli(str(one))
    assert rc == 0, err
    lines = out.splitlines()
    assert lines[0] == "only" and lines[1] == ">SS_cons" and lines[3] == "> only" and lines[4] == "GGGAAACCCUUUAGGGCCC"
    recs = synth.family_set(6, 60, seed=8)
    fa = tmp_path / "f.fa"
    fa.write_text(synth.to_fasta(recs))
    rc, out, err = run_cli("-r", "3", str(fa))
    assert rc == 0, err
    rows = out.splitlines()[3:]
    for (name, seq), hdr, row in zip(recs, rows[0::2], rows[1::2]):
        assert hdr == "> " + name and row.replace("-", "") == seq and len(row) == len(rows[1])


def test_errors_follow_the_reference_convention(tmp_path):
    path = os.path.join(G, "RF00005_0.fa")
    for args, msg in ((["-s", "Boltzmann", path], "ViennaRNA"), (["-a", "Nope", path], "Unknown alignment model: Nope"),
                      (["--ipknot", path], "ILP"), ([str(tmp_path / "missing.fa")], "missing.fa")):
        rc, out, err = run_cli(*args)
        assert rc == 1 and msg in err and out == "", (args, err)   # message on stderr, EXIT_FAILURE (dafs.cpp:1893-1910)


def test_devices_flag_shards_phase1_across_processes(tmp_path):
    """dafs --devices: phase 1 as one process per listed device (dafs_hip_phase1_sharded).  `--devices 0`
    is one rank whose exchanges go through RCCL (ncclAllGather on a communicator of one); `--devices 0,0,0` is three
    processes on the one GPU of this box, which RCCL refuses, so their shards travel through the host staging area -- the
    same library path, the same process structure.  Every variant must print what the single-process run prints, and the
    aux files written from the gathered stores must be the single-process ones byte for byte."""
    path = os.path.join(G, "RF00005_0.fa")
    fa = str(tmp_path / "fam.fa")
    with open(fa, "w") as f:
        for n, s in synth.family_set(11, 70, seed=7):
            f.write(">%s\n%s\n" % (n, s))
    for inp, flags in ((path, []), (path, ["-a", "CONTRAlign", "-p", "0.3", "-q", "0.2"]), (fa, ["-m", "80"])):
        rc, want, err = run_cli(*flags, "--save-fold-aux", str(tmp_path / "f0"), "--save-align-aux", str(tmp_path / "a0"), inp)
        assert rc == 0, err
        for devs in ("0", "0,0", "0,0,0"):
            rc, out, err = run_cli(*flags, "--devices", devs, "--save-fold-aux", str(tmp_path / "f1"), "--save-align-aux", str(tmp_path / "a1"), inp)
            assert rc == 0, (devs, err)
            assert out == want, devs
            assert open(str(tmp_path / "f1")).read() == open(str(tmp_path / "f0")).read(), devs
            assert open(str(tmp_path / "a1")).read() == open(str(tmp_path / "a0")).read(), devs
    # more ranks than pairs or sequences: the idle ranks still take part in every exchange
    two = str(tmp_path / "two.fa")
    with open(two, "w") as f:
        f.write(">a\nGGGAAACCCUUAGC\n>b\nGGCAAAGCCUAGC\n")
    rc, want, err = run_cli(two)
    assert rc == 0, err
    rc, out, err = run_cli("--devices", "0,0,0", two)
    assert rc == 0, err
    assert out == want
    # what is not sharded says so; a rank that fails takes the run down instead of hanging it
    rc, out, err = run_cli("--devices", "0,0", "-f", "0.1", path)
    assert rc != 0 and "--devices" in err
    rc, out, err = run_cli("--devices", "0,99", path)
    assert rc != 0


def _bits(h):
    return struct.unpack("<f", struct.pack("<I", int(h, 16)))[0]


def test_plugin_classes_against_oracle(oracle):
    import test_oracle_cpu as t
    path = os.path.join(G, "RF00005_0.fa")
    seqs = [s for _, s in oracle.fasta(path)]
    r = subprocess.run([SELFTEST, path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = {}
    other = {}
    for line in r.stdout.splitlines():
        f = line.split()
        if f[0] in ("BP", "BP1", "BPC", "MPP", "MPC", "MPP1", "MPC1"):
            rows.setdefault((f[0], int(f[1]), int(f[2])), []).append((int(f[3]), int(f[4]), int(f[5], 16)))
        else:
            other[f[0]] = f[1:]

    def want_rows(rp, col, val):
        out = []
        for i in range(len(rp) - 1):
            for k in range(rp[i], rp[i + 1]):
                out.append((i, int(col[k]), int(np.float32(val[k]).view(np.uint32))))
        return out

    def fold_rows(s, cons=None):
        L = len(s)
        rp = np.zeros(L + 1, np.uint32); col = np.zeros(L * L + 1, np.uint32); val = np.zeros(L * L + 1, np.float32)
        n = oracle.lib.orc_fold_calculate(s.encode(), L, None if cons is None else cons.encode(), 0.01, rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        return want_rows(rp, col[:n], val[:n])

    for x, s in enumerate(seqs):
        assert rows.get(("BP", x, 0), []) == fold_rows(s), x
    assert rows[("BP1", 0, 0)] == fold_rows(seqs[0])
    cons = "(" + "?" * 2 + "." + "?" * (len(seqs[0]) - 5) + ")"
    assert rows[("BPC", 0, 0)] == fold_rows(seqs[0], cons)
    for tag, model in (("MPP", 0), ("MPC", 1)):
        for i in range(len(seqs)):
            assert rows[(tag, i, i)] == [(k, k, 0x3F800000) for k in range(len(seqs[i]))]
            for j in range(i + 1, len(seqs)):
                assert rows.get((tag, i, j), []) == want_rows(*oracle.align_calculate(seqs[i], seqs[j], 0.01, model)), (tag, i, j)
        assert rows[(tag + "1", 0, 1)] == want_rows(*oracle.align_calculate(seqs[0], seqs[1], 0.01, model))
    # decoders, on the same dense matrices the self-test builds
    rp, col, val = oracle.align_calculate(seqs[0], seqs[1], 0.01, 0)
    L1, L2 = len(seqs[0]), len(seqs[1])
    p = np.zeros((L1, L2), np.float32); q = np.zeros((L1, L2), np.float32)
    for i in range(L1):
        for k in range(rp[i], rp[i + 1]):
            p[i, col[k]] = val[k]; q[i, col[k]] = np.float32(0.125) * np.float32((i + int(col[k])) % 3)
    for tag, qq in (("NWQ", q), ("NW", None)):
        s, al = oracle.nw(p, qq, 0.01)
        assert int(other[tag][0], 16) == int(np.float32(s).view(np.uint32))
        assert [int(v) for v in other[tag][1:]] == [int(np.int32(v)) for v in al.view(np.int32)]
    frows = fold_rows(seqs[0])
    L = len(seqs[0])
    p = np.zeros((L, L), np.float32); q = np.zeros((L, L), np.float32)
    for i, j, b in frows:
        p[i, j] = np.uint32(b).view(np.float32); q[i, j] = np.float32(0.25) * np.float32((i * 7 + j) % 4) - np.float32(0.25)
    s, ss = oracle.nussinov(p, q, 0.2, 4.0)
    assert int(other["NUQ"][0], 16) == int(np.float32(s).view(np.uint32))
    assert [int(v) for v in other["NUQ"][1:]] == [int(v) for v in ss.view(np.int32)]
    s, ss = oracle.nussinov(p, None, 0.2)
    import ctypes as C
    buf = C.create_string_buffer(L + 1)
    oracle.lib.orc_make_brackets(L, ss.ctypes.data, buf)
    assert int(other["NU"][0], 16) == int(np.float32(s).view(np.uint32)) and other["NU"][1] == buf.value.decode()
