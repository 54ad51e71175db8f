"""Loader + numpy wrappers for the CHECKERS: oracle/liboracle.so (our CPU restatement) and
oracle/_ref/libdafs_ref.so (the reference's own sources compiled by oracle/Makefile).
Test infrastructure only -- nothing under dafs_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NONE = 0xFFFFFFFF


def _build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "all"], check=True, capture_output=True)


class OrcParams(C.Structure):
    _fields_ = [("align_model", C.c_int), ("fold_model", C.c_int), ("w", C.c_float), ("eta0", C.c_float),
                ("t_max", C.c_uint), ("w_pct_a", C.c_float), ("w_pct_s", C.c_float), ("th_a", C.c_float),
                ("th_s", C.c_float), ("th_s1", C.c_float), ("force_iters", C.c_int), ("w_pct_f", C.c_float),
                ("bp_update", C.c_int), ("bp_update1", C.c_int)]


class OrcCsr(C.Structure):
    _fields_ = [("nrow", C.c_uint32), ("rowptr", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)),
                ("val", C.POINTER(C.c_float))]


def _csr_to_np(c):
    n = c.nrow
    rp = np.ctypeslib.as_array(c.rowptr, (n + 1,)).copy()
    nnz = int(rp[-1])
    if nnz == 0:
        return rp, np.zeros(0, np.uint32), np.zeros(0, np.float32)
    return rp, np.ctypeslib.as_array(c.col, (nnz,)).copy(), np.ctypeslib.as_array(c.val, (nnz,)).copy()


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_probcons_posterior.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_float, C.c_void_p]
        L.orc_contralign_posterior.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_float, C.c_void_p]
        L.orc_align_calculate.argtypes = [C.c_int, C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_float] + [C.c_void_p] * 3
        L.orc_contrafold_posterior.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_void_p]
        L.orc_fold_calculate.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_float] + [C.c_void_p] * 3
        L.orc_nussinov_decode.restype = C.c_float
        L.orc_nussinov_decode.argtypes = [C.c_float, C.c_float, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_nw_envelope.argtypes = [C.c_float, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        L.orc_nw_decode.restype = C.c_float
        L.orc_nw_decode.argtypes = [C.c_float, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_similarity_score.restype = C.c_float
        L.orc_similarity_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
        L.orc_make_brackets.argtypes = [C.c_uint, C.c_void_p, C.c_char_p]
        L.orc_pipeline_new.restype = C.c_void_p
        L.orc_pipeline_new.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.orc_pipeline_free.argtypes = [C.c_void_p]
        L.orc_pipeline_set_bp.argtypes = [C.c_void_p, C.c_uint] + [C.c_void_p] * 3
        L.orc_pipeline_set_mp.argtypes = [C.c_void_p, C.c_uint, C.c_uint] + [C.c_void_p] * 3
        L.orc_pipeline_phase1.argtypes = [C.c_void_p]
        L.orc_pipeline_phase2.argtypes = [C.c_void_p]
        L.orc_pipeline_output.restype = C.c_char_p
        L.orc_pipeline_output.argtypes = [C.c_void_p]
        L.orc_pipeline_mp.restype = C.POINTER(OrcCsr)
        L.orc_pipeline_mp.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
        L.orc_pipeline_bp.restype = C.POINTER(OrcCsr)
        L.orc_pipeline_bp.argtypes = [C.c_void_p, C.c_uint]
        L.orc_pipeline_sim.restype = C.POINTER(C.c_float)
        L.orc_pipeline_sim.argtypes = [C.c_void_p]
        L.orc_pipeline_tree.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.orc_pipeline_dd_log.restype = C.c_uint
        L.orc_pipeline_dd_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        L.orc_pipeline_seconds.restype = C.c_double
        L.orc_pipeline_seconds.argtypes = [C.c_void_p, C.c_int]
        L.orc_fasta_load.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        L.orc_fasta_free.argtypes = [C.c_int, C.c_void_p, C.c_void_p]

    # --- single-call helpers ---
    def probcons_posterior(self, s1, s2, th=0.0):
        a, b = s1.encode(), s2.encode()
        out = np.zeros((len(a) + 1) * (len(b) + 1), np.float32)
        rc = self.lib.orc_probcons_posterior(a, len(a), b, len(b), th, out.ctypes.data)
        assert rc >= 0
        return out.reshape(len(a) + 1, len(b) + 1)

    def align_calculate(self, s1, s2, th=0.01, model=0):
        a, b = s1.encode(), s2.encode()
        rp = np.zeros(len(a) + 1, np.uint32)
        col = np.zeros(len(a) * len(b), np.uint32)
        val = np.zeros(len(a) * len(b), np.float32)
        n = self.lib.orc_align_calculate(model, a, len(a), b, len(b), th, rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        assert n >= 0, n
        return rp, col[:n].copy(), val[:n].copy()

    def similarity(self, rp, col, val, L1, L2):
        rp = np.ascontiguousarray(rp, np.uint32); col = np.ascontiguousarray(col, np.uint32)
        val = np.ascontiguousarray(val, np.float32)
        return np.float32(self.lib.orc_similarity_score(rp.ctypes.data, col.ctypes.data, val.ctypes.data, L1, L2))

    def nussinov(self, p, q, th, w=0.0):
        p = np.ascontiguousarray(p, np.float32)
        L = p.shape[0]
        ss = np.zeros(L, np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        s = self.lib.orc_nussinov_decode(th, w, L, p.ctypes.data, None if qp is None else qp.ctypes.data, ss.ctypes.data)
        return np.float32(s), ss

    def nussinov_dense(self, p, q, th, w=0.0):
        p = np.ascontiguousarray(p, np.float32)
        L = p.shape[0]
        ss = np.zeros(L, np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        self.lib.orc_nussinov_dense_decode.restype = C.c_float
        self.lib.orc_nussinov_dense_decode.argtypes = [C.c_float, C.c_float, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        s = self.lib.orc_nussinov_dense_decode(th, w, L, p.ctypes.data, None if qp is None else qp.ctypes.data, ss.ctypes.data)
        return np.float32(s), ss

    def nw_dense(self, p, q, th):
        p = np.ascontiguousarray(p, np.float32)
        al = np.zeros(p.shape[0], np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        self.lib.orc_nw_dense_decode.restype = C.c_float
        self.lib.orc_nw_dense_decode.argtypes = [C.c_float, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        s = self.lib.orc_nw_dense_decode(th, p.shape[0], p.shape[1], p.ctypes.data, None if qp is None else qp.ctypes.data, al.ctypes.data)
        return np.float32(s), al

    def nw_envelope(self, p, th):
        p = np.ascontiguousarray(p, np.float32)
        env = np.zeros(2 * (p.shape[0] + 1), np.uint32)
        self.lib.orc_nw_envelope(th, p.shape[0], p.shape[1], p.ctypes.data, env.ctypes.data)
        return env

    def nw(self, p, q, th, env=None):
        p = np.ascontiguousarray(p, np.float32)
        if env is None:
            env = self.nw_envelope(p, th)
        al = np.zeros(p.shape[0], np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        s = self.lib.orc_nw_decode(th, p.shape[0], p.shape[1], p.ctypes.data, None if qp is None else qp.ctypes.data,
                                   env.ctypes.data, al.ctypes.data)
        return np.float32(s), al

    def fasta(self, path):
        names = C.POINTER(C.c_char_p)(); seqs = C.POINTER(C.c_char_p)()
        n = self.lib.orc_fasta_load(path.encode(), C.byref(names), C.byref(seqs))
        assert n >= 0, path
        out = [(names[i].decode(), seqs[i].decode()) for i in range(n)]
        self.lib.orc_fasta_free(n, names, seqs)
        return out

    def params(self, **kw):
        p = OrcParams()
        self.lib.orc_params_default(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def pipeline(self, names, seqs, prm=None, bp=None, mp=None):
        """mp (with prm.align_model == 2): function (x, y) -> (rowptr, col, val) of mp[x][y], x < y"""
        return Pipeline(self, names, seqs, prm or self.params(), bp, mp)


class Pipeline:
    def __init__(self, orc, names, seqs, prm, bp, mp=None):
        self.o = orc
        self.N = len(seqs)
        self.lens = [len(s) for s in seqs]
        na = (C.c_char_p * self.N)(*[n.encode() for n in names])
        sa = (C.c_char_p * self.N)(*[s.encode() for s in seqs])
        self.h = orc.lib.orc_pipeline_new(C.byref(prm), self.N, na, sa)
        if bp is not None:
            for x, (rp, col, val) in enumerate(bp):
                rp = np.ascontiguousarray(rp, np.uint32); col = np.ascontiguousarray(col, np.uint32)
                val = np.ascontiguousarray(val, np.float32)
                orc.lib.orc_pipeline_set_bp(self.h, x, rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        if mp is not None:
            for x in range(self.N):
                for y in range(x + 1, self.N):
                    self.set_mp(x, y, *mp(x, y))

    def set_mp(self, x, y, rp, col, val):
        rp = np.ascontiguousarray(rp, np.uint32); col = np.ascontiguousarray(col, np.uint32)
        val = np.ascontiguousarray(val, np.float32)
        assert len(rp) == self.lens[x] + 1 and len(col) == len(val) == int(rp[-1])
        self.o.lib.orc_pipeline_set_mp(self.h, x, y, rp.ctypes.data, col.ctypes.data, val.ctypes.data)

    def phase1(self):
        rc = self.o.lib.orc_pipeline_phase1(self.h); assert rc == 0, rc

    def phase2(self):
        rc = self.o.lib.orc_pipeline_phase2(self.h); assert rc == 0, rc

    def output(self):
        return self.o.lib.orc_pipeline_output(self.h).decode()

    def mp(self, x, y):
        return _csr_to_np(self.o.lib.orc_pipeline_mp(self.h, x, y).contents)

    def bp(self, x):
        return _csr_to_np(self.o.lib.orc_pipeline_bp(self.h, x).contents)

    def sim(self):
        return np.ctypeslib.as_array(self.o.lib.orc_pipeline_sim(self.h), (self.N * self.N,)).reshape(self.N, self.N).copy()

    def tree(self):
        T = 2 * self.N - 1
        s = np.zeros(T, np.float32); l = np.zeros(T, np.uint32); r = np.zeros(T, np.uint32)
        self.o.lib.orc_pipeline_tree(self.h, s.ctypes.data, l.ctypes.data, r.ctypes.data)
        return s, l, r

    def dd_log(self):
        it = np.zeros(self.N + 1, np.uint32); vi = np.zeros(self.N + 1, np.uint32)
        k = self.o.lib.orc_pipeline_dd_log(self.h, it.ctypes.data, vi.ctypes.data, self.N + 1)
        return it[:k], vi[:k]

    def seconds(self):
        return [self.o.lib.orc_pipeline_seconds(self.h, i) for i in range(4)]

    def close(self):
        if self.h:
            self.o.lib.orc_pipeline_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _phase1_worker(job):
    """one worker of parallel_oracle_run: base-pairing rows of its sequences and matching rows of its pairs (CPU only)"""
    seqs, xs, pairs, model, th = job
    orc = load_oracle()
    bp = {}
    for x in xs:
        L = len(seqs[x])
        rp = np.zeros(L + 1, np.uint32); col = np.zeros(L * (L + 1) // 2 + 1, np.uint32); val = np.zeros(L * (L + 1) // 2 + 1, np.float32)
        n = orc.lib.orc_fold_calculate(seqs[x].encode(), L, None, C.c_float(0.01), rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        assert n >= 0
        bp[x] = (rp, col[:n].copy(), val[:n].copy())
    mp = {(x, y): orc.align_calculate(seqs[x], seqs[y], th, model) for x, y in pairs}
    return bp, mp


def _parallel_oracle_main():
    """child process of parallel_oracle_run (python oracle_lib.py --run): job as JSON on stdin, result as JSON on stdout"""
    import json
    import multiprocessing as mp
    import os
    import sys
    job = json.load(sys.stdin)
    names, seqs, params = job["names"], job["seqs"], job["params"]
    model = params.pop("align_model", 0)
    th = params.get("th_a", 0.01)
    n = len(seqs)
    workers = job.get("workers") or min(len(os.sched_getaffinity(0)), 16)
    pairs = [(x, y) for x in range(n) for y in range(x + 1, n)]
    jobs = [(seqs, list(range(w, n, workers)), pairs[w::workers], model, th) for w in range(workers)]
    with mp.get_context("fork").Pool(workers) as pool:  # this process never touches a GPU
        parts = pool.map(_phase1_worker, jobs)
    bp, rows = {}, {}
    for b, m in parts:
        bp.update(b); rows.update(m)
    orc = load_oracle()
    pl = orc.pipeline(names, seqs, orc.params(fold_model=1, align_model=2, **params), bp=[bp[x] for x in range(n)], mp=lambda x, y: rows[(x, y)])
    pl.phase1(); pl.phase2()
    it, vi = pl.dd_log()
    json.dump({"output": pl.output(), "iterations": [int(v) for v in it], "violated": [int(v) for v in vi], "seconds": pl.seconds()}, sys.stdout)
    pl.close()


def parallel_oracle_run(names, seqs, workers=None, **params):
    """The oracle's whole run with its phase-1 models computed process-parallel (the reference has no threading; this only
    shortens the checker): CONTRAfold rows and matching rows of the chosen align_model come from worker processes and enter
    the pipeline through its --fold-aux / --align-aux paths (orc_pipeline_set_bp / _set_mp); consistency transforms, tree
    and progressive phase run on one core.  All of it in a child process of its own (the caller may hold a GPU context).
    Returns (output text, (iterations, violated) log)."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--run"], input=json.dumps({"names": names, "seqs": seqs, "params": params, "workers": workers}),
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle run failed:\n" + r.stderr[-2000:])
    d = json.loads(r.stdout)
    return d["output"], (np.array(d["iterations"], np.uint32), np.array(d["violated"], np.uint32))


def load_oracle():
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path) or os.path.exists("/root/reference/src/dafs.cpp"):
        try:
            _build()
        except Exception:
            if not os.path.exists(path):
                raise
    return Oracle(C.CDLL(path))


class Ref:
    """oracle/_ref: the reference's own code."""

    def __init__(self, lib):
        self.lib = lib
        lib.ref_probcons_posterior.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_void_p]
        lib.ref_contralign_posterior.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_void_p]
        lib.ref_align_calculate.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_float] + [C.c_void_p] * 3
        lib.ref_contrafold_posterior.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p]
        lib.ref_nussinov_decode.restype = C.c_float
        lib.ref_nussinov_decode.argtypes = [C.c_float, C.c_float, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_nussinov_decode_final.restype = C.c_float
        lib.ref_nussinov_decode_final.argtypes = [C.c_float, C.c_uint, C.c_void_p, C.c_void_p, C.c_char_p]
        lib.ref_nw_decode.restype = C.c_float
        lib.ref_nw_decode.argtypes = [C.c_float, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_fasta_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]

    def probcons_posterior(self, s1, s2, th=0.0):
        a, b = s1.encode(), s2.encode()
        out = np.zeros((len(a) + 1) * (len(b) + 1), np.float32)
        self.lib.ref_probcons_posterior(a, b, th, out.ctypes.data)
        return out.reshape(len(a) + 1, len(b) + 1)

    def contralign_posterior(self, s1, s2, th=0.0):
        a, b = s1.encode(), s2.encode()
        out = np.zeros((len(a) + 1) * (len(b) + 1), np.float32)
        self.lib.ref_contralign_posterior(a, b, th, out.ctypes.data)
        return out.reshape(len(a) + 1, len(b) + 1)

    def align_calculate(self, s1, s2, th=0.01, model=0):
        a, b = s1.encode(), s2.encode()
        rp = np.zeros(len(a) + 1, np.uint32)
        col = np.zeros(len(a) * len(b), np.uint32); val = np.zeros(len(a) * len(b), np.float32)
        n = self.lib.ref_align_calculate(model, a, b, th, rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        return rp, col[:n].copy(), val[:n].copy()

    def contrafold_posterior(self, s, constraint=None):
        a = s.encode()
        L = len(a)
        out = np.zeros((L + 1) * (L + 2) // 2, np.float32)
        self.lib.ref_contrafold_posterior(a, None if constraint is None else constraint.encode(), out.ctypes.data)
        return out

    def nussinov(self, p, q, th, w=0.0):
        p = np.ascontiguousarray(p, np.float32); L = p.shape[0]
        ss = np.zeros(L, np.uint32)
        if q is None:
            buf = C.create_string_buffer(L + 1)
            s = self.lib.ref_nussinov_decode_final(th, L, p.ctypes.data, ss.ctypes.data, buf)
            return np.float32(s), ss, buf.value.decode()
        q = np.ascontiguousarray(q, np.float32)
        s = self.lib.ref_nussinov_decode(th, w, L, p.ctypes.data, q.ctypes.data, ss.ctypes.data)
        return np.float32(s), ss

    def nw(self, p, q, th):
        p = np.ascontiguousarray(p, np.float32)
        al = np.zeros(p.shape[0], np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        s = self.lib.ref_nw_decode(th, p.shape[0], p.shape[1], p.ctypes.data, None if qp is None else qp.ctypes.data, al.ctypes.data)
        return np.float32(s), al

    def nussinov_dense(self, p, q, th, w=0.0):
        p = np.ascontiguousarray(p, np.float32); L = p.shape[0]
        ss = np.zeros(L, np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        self.lib.ref_nussinov_dense_decode.restype = C.c_float
        self.lib.ref_nussinov_dense_decode.argtypes = [C.c_float, C.c_float, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        s = self.lib.ref_nussinov_dense_decode(th, w, L, p.ctypes.data, None if qp is None else qp.ctypes.data, ss.ctypes.data)
        return np.float32(s), ss

    def nw_dense(self, p, q, th):
        p = np.ascontiguousarray(p, np.float32)
        al = np.zeros(p.shape[0], np.uint32)
        qp = None if q is None else np.ascontiguousarray(q, np.float32)
        self.lib.ref_nw_dense_decode.restype = C.c_float
        self.lib.ref_nw_dense_decode.argtypes = [C.c_float, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        s = self.lib.ref_nw_dense_decode(th, p.shape[0], p.shape[1], p.ctypes.data, None if qp is None else qp.ctypes.data, al.ctypes.data)
        return np.float32(s), al

    def auxalign_load(self, path, seqs):
        """the reference's own --align-aux reader (AUXAlign::calculate, src/align.cpp:204-246): (nnz, rowptr, col, val) of
        every pair x < y in row-major order, rows of mp[x][y] (the layout of Context.set_mp)"""
        n = len(seqs)
        arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
        npairs = n * (n - 1) // 2
        lens = [len(s) for s in seqs]
        rp_cap = sum(lens[x] + 1 for x in range(n) for _ in range(x + 1, n))
        ent_cap = sum(lens[x] * lens[y] for x in range(n) for y in range(x + 1, n))
        nnz = np.zeros(npairs, np.uint32); rowptr = np.zeros(rp_cap, np.uint32)
        col = np.zeros(ent_cap, np.uint32); val = np.zeros(ent_cap, np.float32)
        self.lib.ref_auxalign_load.restype = C.c_long
        self.lib.ref_auxalign_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        e = self.lib.ref_auxalign_load(path.encode(), n, arr, nnz.ctypes.data, rowptr.ctypes.data, rp_cap, col.ctypes.data, val.ctypes.data, ent_cap)
        assert e >= 0
        return nnz, rowptr, col[:e], val[:e]

    def fasta(self, path):
        nb = C.create_string_buffer(1 << 20); sb = C.create_string_buffer(1 << 22)
        n = self.lib.ref_fasta_load(path.encode(), nb, len(nb), sb, len(sb))
        assert n >= 0
        return list(zip(nb.value.decode().split("\n")[:n], sb.value.decode().split("\n")[:n]))


def load_ref():
    path = os.path.join(ORACLE_DIR, "_ref", "libdafs_ref.so")
    if not os.path.exists(path):
        return None
    return Ref(C.CDLL(path))


if __name__ == "__main__":
    import sys
    if "--run" in sys.argv:
        _parallel_oracle_main()
