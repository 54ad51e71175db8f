"""ctypes binding of include/dafs_hip.h (the same entry points a cgo/JNI/C++ shim would bind)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAFS_HIP_LIB") or os.path.join(_HERE, "libdafs_hip.so")  # DAFS_HIP_LIB: tuning builds

NONE = 0xFFFFFFFF
ALIGN_PROBCONS, ALIGN_CONTRALIGN = 0, 1
E_OVERFLOW = -5


class DafsHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise DafsHipError(
            "libdafs_hip.so is not built (%s). Run `python -m dafs_amd.build`; there is no CPU fallback." % LIB_PATH)
    return C.CDLL(LIB_PATH)


lib = _load()

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)


class PairTask(C.Structure):
    _fields_ = [("off1", C.c_uint32), ("len1", C.c_uint32), ("off2", C.c_uint32), ("len2", C.c_uint32)]


class PairhmmPlan(C.Structure):
    _fields_ = [("group", C.c_uint32), ("width", C.c_uint32), ("nwaves", C.c_uint32), ("slab_steps", C.c_uint32),
                ("scratch_bytes", C.c_uint64)]


class Pairhmm3Model(C.Structure):
    _fields_ = [("init", C.c_float * 3), ("trans", (C.c_float * 3) * 3), ("match", (C.c_float * 8) * 7),
                ("ins", C.c_float * 8)]


class Pairhmm3Args(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("tasks", C.c_void_p), ("ntasks", C.c_uint32), ("th", C.c_float),
                ("scratch", C.c_void_p), ("queue", C.c_void_p), ("rp_off", C.c_void_p), ("rowptr_pool", C.c_void_p),
                ("ent_col", C.c_void_p), ("ent_val", C.c_void_p), ("pool_top", C.c_void_p), ("pool_cap", C.c_uint64),
                ("pair_off", C.c_void_p), ("pair_nnz", C.c_void_p), ("sim", C.c_void_p), ("status", C.c_void_p),
                ("model", Pairhmm3Model)]


class Pairhmm5Model(C.Structure):
    _fields_ = [("match", (C.c_float * 5) * 5), ("insert", C.c_float * 5), ("single", C.c_float * 5),
                ("pair", (C.c_float * 5) * 5)]


class Pairhmm5Args(C.Structure):
    _fields_ = Pairhmm3Args._fields_[:-1] + [("model", Pairhmm5Model)]


def _sig(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


_strerror = _sig("dafs_hip_strerror", C.c_char_p, [C.c_int])
_last_error = _sig("dafs_hip_last_error", C.c_char_p, [])
_create = _sig("dafs_hip_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)])
_destroy = _sig("dafs_hip_destroy", None, [C.c_void_p])
_set_sequences = _sig("dafs_hip_set_sequences", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), u32p])
_align_posteriors = _sig("dafs_hip_align_posteriors", C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint64])
_align_result_size = _sig("dafs_hip_align_result_size", C.c_int,
                          [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
_align_fetch = _sig("dafs_hip_align_fetch", C.c_int, [C.c_void_p] + [C.c_void_p] * 7)
_mp_result_size = _sig("dafs_hip_mp_result_size", C.c_int,
                      [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
_mp_fetch = _sig("dafs_hip_mp_fetch", C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6)
_get_sim = _sig("dafs_hip_get_sim", C.c_int, [C.c_void_p, C.c_void_p])
_set_bp = _sig("dafs_hip_set_bp", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_bp_result_size = _sig("dafs_hip_bp_result_size", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
_bp_fetch = _sig("dafs_hip_bp_fetch", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
_fold_posteriors = _sig("dafs_hip_fold_posteriors", C.c_int, [C.c_void_p, C.c_int, C.c_float])
_fold_posterior_dense = _sig("dafs_hip_fold_posterior_dense", C.c_int,
                             [C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_void_p, C.POINTER(C.c_float)])
_consistency = _sig("dafs_hip_consistency", C.c_int, [C.c_void_p, C.c_float, C.c_float])
_consistency_match = _sig("dafs_hip_consistency_match", C.c_int, [C.c_void_p, C.c_float])
_consistency_bp = _sig("dafs_hip_consistency_bp", C.c_int, [C.c_void_p, C.c_float])
_fourway_consistency = _sig("dafs_hip_fourway_consistency", C.c_int, [C.c_void_p, C.c_float])
_consistency_match_range = _sig("dafs_hip_consistency_match_range", C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.c_uint64])
_mp_install = _sig("dafs_hip_mp_install", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_fold_begin = _sig("dafs_hip_fold_posteriors_begin", C.c_int, [C.c_void_p, C.c_int, C.c_float])
_fold_end = _sig("dafs_hip_fold_posteriors_end", C.c_int, [C.c_void_p])


class NodeInput(C.Structure):
    _fields_ = [("n1", C.c_uint32), ("n2", C.c_uint32), ("len1", C.c_uint32), ("len2", C.c_uint32),
                ("seq1", C.c_void_p), ("seq2", C.c_void_p), ("mask1", C.c_void_p), ("mask2", C.c_void_p),
                ("p_x", C.c_void_p), ("p_y", C.c_void_p)]


class NodeOutput(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("score", C.c_float),
                ("ncbp", C.c_uint32), ("iterations", C.c_uint32), ("violated", C.c_uint32)]


class DDParams(C.Structure):
    _fields_ = [("w", C.c_float), ("eta0", C.c_float), ("th_a", C.c_float), ("th_s", C.c_float),
                ("t_max", C.c_uint32), ("force_iters", C.c_int), ("skip_uncoupled_folds", C.c_int)]


_nussinov_decode = _sig("dafs_hip_nussinov_decode", C.c_int,
                        [C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)])
_nw_envelope = _sig("dafs_hip_nw_envelope", C.c_int, [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p])
_nw_decode = _sig("dafs_hip_nw_decode", C.c_int,
                  [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)])
_nussinov_decode_dense = _sig("dafs_hip_nussinov_decode_dense", C.c_int,
                              [C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)])
_nw_decode_dense = _sig("dafs_hip_nw_decode_dense", C.c_int,
                        [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)])
_make_brackets = _sig("dafs_hip_make_brackets", None, [C.c_uint32, C.c_void_p, C.c_char_p])
_dd_default_params = _sig("dafs_hip_dd_default_params", None, [C.POINTER(DDParams)])
_solve_nodes = _sig("dafs_hip_solve_nodes", C.c_int,
                    [C.c_void_p, C.c_uint32, C.POINTER(NodeInput), C.POINTER(DDParams), C.POINTER(NodeOutput)])
_build_tree = _sig("dafs_host_build_tree", C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_set_mp = _sig("dafs_hip_set_mp", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_nodes_open = _sig("dafs_hip_nodes_open", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(NodeInput), C.POINTER(DDParams), C.c_void_p])
_nodes_advance = _sig("dafs_hip_nodes_advance", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(DDParams), C.c_uint32, C.c_void_p])
_nodes_round = _sig("dafs_hip_nodes_round", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(NodeInput), C.c_void_p, C.c_uint32, C.c_void_p,
                                                      C.POINTER(DDParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p])
_nodes_result = _sig("dafs_hip_nodes_result", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(NodeOutput)])
_nodes_close = _sig("dafs_hip_nodes_close", C.c_int, [C.c_void_p])
_nodes_memory = _sig("dafs_hip_nodes_memory", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
_nodes_demotions = _sig("dafs_hip_nodes_demotions", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)])
_update_basepairing = _sig("dafs_hip_update_basepairing", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_consensus_structure = _sig("dafs_hip_consensus_structure", C.c_int,
                            [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                             C.POINTER(C.c_float), C.c_void_p])
_mp_export_dev = _sig("dafs_hip_mp_export_dev", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
_mp_install_dev = _sig("dafs_hip_mp_install_dev", C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_uint64])
_bp_export_dev = _sig("dafs_hip_bp_export_dev", C.c_int, [C.c_void_p] + [C.c_void_p] * 3 + [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
_set_bp_dev = _sig("dafs_hip_set_bp_dev", C.c_int, [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint64])


class StageTime(C.Structure):
    _fields_ = [("kernel", C.c_char_p), ("ms", C.c_double), ("longest_ms", C.c_double), ("launches", C.c_uint32)]


_stage_timing = _sig("dafs_hip_stage_timing", C.c_int, [C.c_void_p, C.c_int])
_stage_report = _sig("dafs_hip_stage_report", C.c_int, [C.c_void_p, C.POINTER(StageTime), C.c_uint32, C.POINTER(C.c_uint32)])
pairhmm_plan = _sig("dafs_hipk_pairhmm_plan", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PairhmmPlan)])
pairhmm3_launch = _sig("dafs_hipk_pairhmm3_launch", C.c_int, [C.POINTER(Pairhmm3Args), C.POINTER(PairhmmPlan), C.c_void_p])
pairhmm3_default_model = _sig("dafs_hip_pairhmm3_default_model", None, [C.POINTER(Pairhmm3Model)])
pairhmm5_plan = _sig("dafs_hipk_pairhmm5_plan", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PairhmmPlan)])
pairhmm5_launch = _sig("dafs_hipk_pairhmm5_launch", C.c_int, [C.POINTER(Pairhmm5Args), C.POINTER(PairhmmPlan), C.c_void_p])
pairhmm5_default_model = _sig("dafs_hip_pairhmm5_default_model", None, [C.POINTER(Pairhmm5Model)])
residue_code = _sig("dafs_hip_residue_code", C.c_uint8, [C.c_char])


def check(rc):
    if rc != 0:
        raise DafsHipError("%s (code %d; hip: %s)" % (_strerror(rc).decode(), rc, _last_error().decode()))


def encode(seq):
    """residue bytes -> class codes (uint8 numpy array)"""
    b = seq.encode() if isinstance(seq, str) else bytes(seq)
    table = np.array([residue_code(bytes([i])) for i in range(256)], dtype=np.uint8)
    return table[np.frombuffer(b, dtype=np.uint8)]


class PairPosteriors:
    """Result of Context.align_posteriors: per pair (in shard order) the sparse matching
    probabilities mp[x][y] (CSR) and mp[y][x] (CSR of the transpose), and sim[x][y]."""

    def __init__(self, pair_x, pair_y, sim, nnz, rowptr, col, val, lens):
        self.pair_x, self.pair_y, self.sim, self.nnz = pair_x, pair_y, sim, nnz
        self._rowptr, self._col, self._val, self._lens = rowptr, col, val, lens
        l1 = lens[pair_x].astype(np.int64) + 1
        l2 = lens[pair_y].astype(np.int64) + 1
        self._rp_off = np.concatenate([[0], np.cumsum(l1 + l2)])[:-1]
        self._ent_off = np.concatenate([[0], np.cumsum(2 * nnz.astype(np.int64))])[:-1]

    def __len__(self):
        return len(self.pair_x)

    def csr(self, p, transposed=False):
        """(rowptr, col, val) of mp[x][y] (or mp[y][x]) for pair number p"""
        l1 = int(self._lens[self.pair_x[p]]) + 1
        l2 = int(self._lens[self.pair_y[p]]) + 1
        r0, e0, n = int(self._rp_off[p]), int(self._ent_off[p]), int(self.nnz[p])
        if not transposed:
            return self._rowptr[r0:r0 + l1], self._col[e0:e0 + n], self._val[e0:e0 + n]
        return self._rowptr[r0 + l1:r0 + l1 + l2], self._col[e0 + n:e0 + 2 * n], self._val[e0 + n:e0 + 2 * n]


class Context:
    """Owns one dafs_hip_ctx (device workspace on one GPU)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(_create(device, C.byref(self._h)))
        self._lens = None
        self.device_index = device

    def close(self):
        if self._h:
            _destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sequences(self, seqs):
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        arr = (C.c_char_p * len(bs))(*bs)
        lens = np.array([len(b) for b in bs], dtype=np.uint32)
        check(_set_sequences(self._h, len(bs), arr, lens.ctypes.data_as(u32p)))
        self._lens = lens

    def align_posteriors(self, model=ALIGN_PROBCONS, th=0.01, pair_begin=0, pair_end=0, fetch=True):
        check(_align_posteriors(self._h, model, th, pair_begin, pair_end))
        if not fetch:
            return None
        npairs, nnz, nrp = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_align_result_size(self._h, C.byref(npairs), C.byref(nnz), C.byref(nrp)))
        n = npairs.value
        px = np.zeros(n, np.uint32); py = np.zeros(n, np.uint32)
        sim = np.zeros(n, np.float32); cnt = np.zeros(n, np.uint32)
        rowptr = np.zeros(nrp.value, np.uint32)
        col = np.zeros(2 * nnz.value, np.uint32); val = np.zeros(2 * nnz.value, np.float32)
        check(_align_fetch(self._h, px.ctypes.data, py.ctypes.data, sim.ctypes.data, cnt.ctypes.data,
                           rowptr.ctypes.data, col.ctypes.data, val.ctypes.data))
        return PairPosteriors(px, py, sim, cnt, rowptr, col, val, self._lens)

    def mp(self, relaxed):
        """the matching-probability store (0: model output, 1: after consistency) as PairPosteriors"""
        npairs, nnz, nrp = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_mp_result_size(self._h, relaxed, C.byref(npairs), C.byref(nnz), C.byref(nrp)))
        n = npairs.value
        px = np.zeros(n, np.uint32); py = np.zeros(n, np.uint32); cnt = np.zeros(n, np.uint32)
        rowptr = np.zeros(nrp.value, np.uint32)
        col = np.zeros(2 * nnz.value, np.uint32); val = np.zeros(2 * nnz.value, np.float32)
        check(_mp_fetch(self._h, relaxed, px.ctypes.data, py.ctypes.data, cnt.ctypes.data,
                        rowptr.ctypes.data, col.ctypes.data, val.ctypes.data))
        return PairPosteriors(px, py, None, cnt, rowptr, col, val, self._lens)

    def sim(self):
        n = len(self._lens)
        out = np.zeros((n, n), np.float32)
        check(_get_sim(self._h, out.ctypes.data))
        return out

    def set_bp(self, rows):
        """rows: per sequence (rowptr[len+1], col, val)"""
        rp = np.concatenate([np.asarray(r[0], np.uint32) for r in rows])
        col = np.concatenate([np.asarray(r[1], np.uint32) for r in rows]) if rows else np.zeros(0, np.uint32)
        val = np.concatenate([np.asarray(r[2], np.float32) for r in rows]) if rows else np.zeros(0, np.float32)
        rp = np.ascontiguousarray(rp); col = np.ascontiguousarray(col); val = np.ascontiguousarray(val)
        check(_set_bp(self._h, rp.ctypes.data, col.ctypes.data if len(col) else None, val.ctypes.data if len(val) else None))

    def bp(self, relaxed):
        """per sequence (rowptr, col, val)"""
        nnz, nrp = C.c_uint64(), C.c_uint64()
        check(_bp_result_size(self._h, relaxed, C.byref(nnz), C.byref(nrp)))
        rp = np.zeros(nrp.value, np.uint32); col = np.zeros(nnz.value, np.uint32); val = np.zeros(nnz.value, np.float32)
        check(_bp_fetch(self._h, relaxed, rp.ctypes.data, col.ctypes.data, val.ctypes.data))
        out, r0, e0 = [], 0, 0
        for L in self._lens:
            r = rp[r0:r0 + int(L) + 1]
            n = int(r[-1])
            out.append((r, col[e0:e0 + n], val[e0:e0 + n]))
            r0 += int(L) + 1
            e0 += n
        return out

    def fold_posteriors(self, th=0.01, model=0):
        """CONTRAfold base-pairing posteriors of every sequence -> the un-relaxed bp store"""
        check(_fold_posteriors(self._h, model, th))

    def fold_begin(self, th=0.01, model=0):
        """enqueue the folding kernels on their own stream; fold_end() waits and fills the bp store"""
        check(_fold_begin(self._h, model, th))

    def fold_end(self):
        check(_fold_end(self._h))

    def fold_posterior_dense(self, seq, constraint=None):
        b = seq.encode() if isinstance(seq, str) else bytes(seq)
        L = len(b)
        post = np.zeros((L + 1) * (L + 2) // 2, np.float32)
        logz = C.c_float()
        check(_fold_posterior_dense(self._h, b, L, None if constraint is None else constraint.encode(), post.ctypes.data, C.byref(logz)))
        return post, np.float32(logz.value)

    def consistency(self, w_pct_a=0.25, w_pct_s=0.25):
        check(_consistency(self._h, w_pct_a, w_pct_s))

    def fourway_consistency(self, w_pct_f):
        """DAFS::relax_fourway_consistency (-f): replaces the un-relaxed matching store and recomputes the similarity scores"""
        check(_fourway_consistency(self._h, w_pct_f))

    def consistency_match(self, w_pct_a=0.25):
        check(_consistency_match(self._h, w_pct_a))

    def consistency_match_range(self, w_pct_a, pair_begin, pair_end):
        check(_consistency_match_range(self._h, w_pct_a, pair_begin, pair_end))

    def mp_install(self, relaxed, nnz, rowptr, col, val, sim=None):
        """a whole store from arrays in the layout of mp() / align_posteriors() results (dafs_hip_mp_install)"""
        nnz = np.ascontiguousarray(nnz, np.uint32); rowptr = np.ascontiguousarray(rowptr, np.uint32)
        col = np.ascontiguousarray(col, np.uint32); val = np.ascontiguousarray(val, np.float32)
        sim = None if sim is None else np.ascontiguousarray(sim, np.float32)
        check(_mp_install(self._h, relaxed, nnz.ctypes.data, rowptr.ctypes.data, col.ctypes.data if len(col) else None,
                          val.ctypes.data if len(val) else None, None if sim is None else sim.ctypes.data))

    def consistency_bp(self, w_pct_s=0.25):
        check(_consistency_bp(self._h, w_pct_s))

    # --- decoder plugins ---
    def nussinov(self, p, q, th, w=0.0):
        p = np.ascontiguousarray(p, np.float32)
        L = p.shape[0]
        qq = None if q is None else np.ascontiguousarray(q, np.float32)
        ss = np.zeros(L, np.uint32)
        score = C.c_float()
        check(_nussinov_decode(self._h, th, w, L, p.ctypes.data, None if qq is None else qq.ctypes.data,
                               ss.ctypes.data, C.byref(score)))
        return np.float32(score.value), ss

    def nussinov_dense(self, p, q, th, w=0.0):
        """the dense Nussinov class (q None: the final-decode overload); returns (score, ss)"""
        p = np.ascontiguousarray(p, np.float32)
        q = None if q is None else np.ascontiguousarray(q, np.float32)
        ss = np.zeros(p.shape[0], np.uint32)
        s = C.c_float()
        check(_nussinov_decode_dense(self._h, th, w, p.shape[0], p.ctypes.data, None if q is None else q.ctypes.data, ss.ctypes.data, C.byref(s)))
        return np.float32(s.value), ss

    def nw_dense(self, p, q, th):
        p = np.ascontiguousarray(p, np.float32)
        q = None if q is None else np.ascontiguousarray(q, np.float32)
        al = np.zeros(p.shape[0], np.uint32)
        s = C.c_float()
        check(_nw_decode_dense(self._h, th, p.shape[0], p.shape[1], p.ctypes.data, None if q is None else q.ctypes.data, al.ctypes.data, C.byref(s)))
        return np.float32(s.value), al

    def nw_envelope(self, p, th):
        p = np.ascontiguousarray(p, np.float32)
        env = np.zeros(2 * (p.shape[0] + 1), np.uint32)
        check(_nw_envelope(self._h, th, p.shape[0], p.shape[1], p.ctypes.data, env.ctypes.data))
        return env

    def nw(self, p, q, th, env=None):
        p = np.ascontiguousarray(p, np.float32)
        if env is None:
            env = self.nw_envelope(p, th)
        qq = None if q is None else np.ascontiguousarray(q, np.float32)
        al = np.zeros(p.shape[0], np.uint32)
        score = C.c_float()
        check(_nw_decode(self._h, th, p.shape[0], p.shape[1], p.ctypes.data, None if qq is None else qq.ctypes.data,
                         env.ctypes.data, al.ctypes.data, C.byref(score)))
        return np.float32(score.value), al

    # --- fused node solver ---
    def solve_nodes(self, nodes, prm=None):
        """nodes: list of (seq1, mask1, seq2, mask2) with seq* uint32 arrays and mask* uint8 [n, len].
        Returns list of dicts x, y, z, score, ncbp, iterations, violated."""
        if prm is None:
            prm = dd_params()
        n = len(nodes)
        ins = (NodeInput * n)()
        outs = (NodeOutput * n)()
        keep = []
        for b, (s1, m1, s2, m2) in enumerate(nodes):
            s1 = np.ascontiguousarray(s1, np.uint32); s2 = np.ascontiguousarray(s2, np.uint32)
            m1 = np.ascontiguousarray(m1, np.uint8); m2 = np.ascontiguousarray(m2, np.uint8)
            x = np.zeros(m1.shape[1], np.uint32); y = np.zeros(m2.shape[1], np.uint32); z = np.zeros(m1.shape[1], np.uint32)
            keep.append((s1, s2, m1, m2, x, y, z))
            ins[b].n1, ins[b].n2, ins[b].len1, ins[b].len2 = m1.shape[0], m2.shape[0], m1.shape[1], m2.shape[1]
            ins[b].seq1, ins[b].seq2, ins[b].mask1, ins[b].mask2 = s1.ctypes.data, s2.ctypes.data, m1.ctypes.data, m2.ctypes.data
            outs[b].x, outs[b].y, outs[b].z = x.ctypes.data, y.ctypes.data, z.ctypes.data
        check(_solve_nodes(self._h, n, ins, C.byref(prm), outs))
        return [dict(x=k[4], y=k[5], z=k[6], score=np.float32(outs[b].score), ncbp=outs[b].ncbp,
                     iterations=outs[b].iterations, violated=outs[b].violated) for b, k in enumerate(keep)]

    def set_mp(self, nnz, rowptr, col, val):
        """Supplied matching probabilities (--align-aux, or the shards of several GPUs after their all-gather): per
        pair x<y in row-major order nnz[p], then len[x]+1 relative row pointers, then the (col, val) entries."""
        nnz = np.ascontiguousarray(nnz, np.uint32); rowptr = np.ascontiguousarray(rowptr, np.uint32)
        col = np.ascontiguousarray(col, np.uint32); val = np.ascontiguousarray(val, np.float32)
        check(_set_mp(self._h, nnz.ctypes.data, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data))

    # --- resident nodes (no level barrier) ---
    def nodes_open(self, nodes, prm):
        """nodes as in solve_nodes; returns their handles"""
        n = len(nodes)
        ins = (NodeInput * n)()
        keep = []
        for b, (s1, m1, s2, m2) in enumerate(nodes):
            s1 = np.ascontiguousarray(s1, np.uint32); s2 = np.ascontiguousarray(s2, np.uint32)
            m1 = np.ascontiguousarray(m1, np.uint8); m2 = np.ascontiguousarray(m2, np.uint8)
            keep.append((s1, s2, m1, m2))
            ins[b].n1, ins[b].n2, ins[b].len1, ins[b].len2 = m1.shape[0], m2.shape[0], m1.shape[1], m2.shape[1]
            ins[b].seq1, ins[b].seq2, ins[b].mask1, ins[b].mask2 = s1.ctypes.data, s2.ctypes.data, m1.ctypes.data, m2.ctypes.data
        handles = np.zeros(n, np.uint32)
        check(_nodes_open(self._h, n, ins, C.byref(prm), handles.ctypes.data))
        return [int(h) for h in handles], [(k[2].shape[1], k[3].shape[1]) for k in keep]

    def nodes_advance(self, handles, prm, max_iterations):
        """one launch: at most max_iterations more iterations for every listed node; returns the finished flags"""
        h = np.ascontiguousarray(handles, np.uint32)
        fin = np.zeros(len(h), np.uint8)
        check(_nodes_advance(self._h, len(h), h.ctypes.data, C.byref(prm), max_iterations, fin.ctypes.data))
        return fin.astype(bool)

    def nodes_round(self, new_nodes, old_handles, prm, max_iterations, budget_us=0):
        """one round: the open nodes (old_handles) advance while new_nodes (as in solve_nodes) are opened and started beside
        them; at most max_iterations iterations each and, with budget_us, a common stop budget_us microseconds after the
        round began.  Returns (handles of the new nodes, their (len1, len2), finished flags of the old, of the new)."""
        n = len(new_nodes)
        ins = (NodeInput * max(n, 1))()
        keep = []
        for b, node in enumerate(new_nodes):  # (s1, m1, s2, m2) or (s1, m1, s2, m2, p_x, p_y): supplied base-pairing matrices
            s1, m1, s2, m2 = node[:4]
            s1 = np.ascontiguousarray(s1, np.uint32); s2 = np.ascontiguousarray(s2, np.uint32)
            m1 = np.ascontiguousarray(m1, np.uint8); m2 = np.ascontiguousarray(m2, np.uint8)
            px = np.ascontiguousarray(node[4], np.float32) if len(node) > 4 and node[4] is not None else None
            py = np.ascontiguousarray(node[5], np.float32) if len(node) > 5 and node[5] is not None else None
            keep.append((s1, s2, m1, m2, px, py))
            ins[b].n1, ins[b].n2, ins[b].len1, ins[b].len2 = m1.shape[0], m2.shape[0], m1.shape[1], m2.shape[1]
            ins[b].seq1, ins[b].seq2, ins[b].mask1, ins[b].mask2 = s1.ctypes.data, s2.ctypes.data, m1.ctypes.data, m2.ctypes.data
            ins[b].p_x = px.ctypes.data if px is not None else None
            ins[b].p_y = py.ctypes.data if py is not None else None
        nh = np.zeros(max(n, 1), np.uint32)
        h = np.ascontiguousarray(old_handles, np.uint32)
        fo = np.zeros(max(len(h), 1), np.uint8); fn = np.zeros(max(n, 1), np.uint8)
        check(_nodes_round(self._h, n, ins, nh.ctypes.data, len(h), h.ctypes.data if len(h) else None, C.byref(prm), max_iterations,
                           budget_us, fo.ctypes.data, fn.ctypes.data))
        return ([int(x) for x in nh[:n]], [(k[2].shape[1], k[3].shape[1]) for k in keep], fo[:len(h)].astype(bool), fn[:n].astype(bool))

    def nodes_result(self, handle, len1, len2):
        out = NodeOutput()
        x = np.zeros(len1, np.uint32); y = np.zeros(len2, np.uint32); z = np.zeros(len1, np.uint32)
        out.x, out.y, out.z = x.ctypes.data, y.ctypes.data, z.ctypes.data
        check(_nodes_result(self._h, handle, C.byref(out)))
        return dict(x=x, y=y, z=z, score=np.float32(out.score), ncbp=out.ncbp, iterations=out.iterations, violated=out.violated)

    def nodes_close(self):
        check(_nodes_close(self._h))

    def nodes_memory(self):
        """(reserved, in_use, peak) bytes of the resident nodes' device memory"""
        r, u, p = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_nodes_memory(self._h, C.byref(r), C.byref(u), C.byref(p)))
        return r.value, u.value, p.value

    # --- device-resident exchange of the stores (device pointers as integers, e.g. torch.Tensor.data_ptr()) ---
    def mp_sizes(self, relaxed):
        """(pairs, entries = 2 * sum of nnz, row pointers) of a matching store"""
        npairs, nnz, nrp = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_mp_result_size(self._h, relaxed, C.byref(npairs), C.byref(nnz), C.byref(nrp)))
        return npairs.value, 2 * nnz.value, nrp.value

    def mp_export_dev(self, relaxed, first, count, nnz, rowptr, col, val, sim, cap_entries):
        nr, ne = C.c_uint64(), C.c_uint64()
        check(_mp_export_dev(self._h, relaxed, first, count, nnz, rowptr, col, val, sim, cap_entries, C.byref(nr), C.byref(ne)))
        return nr.value, ne.value

    def mp_install_dev(self, relaxed, nnz, rowptr, col, val, sim, n_entries):
        check(_mp_install_dev(self._h, relaxed, nnz, rowptr, col, val, sim, n_entries))

    def bp_sizes(self, relaxed=0):
        nnz, nrp = C.c_uint64(), C.c_uint64()
        check(_bp_result_size(self._h, relaxed, C.byref(nnz), C.byref(nrp)))
        return nnz.value, nrp.value

    def bp_export_dev(self, rowptr, col, val, cap_entries):
        nr, ne = C.c_uint64(), C.c_uint64()
        check(_bp_export_dev(self._h, rowptr, col, val, cap_entries, C.byref(nr), C.byref(ne)))
        return nr.value, ne.value

    def set_bp_dev(self, seq_of_block, rowptr, col, val, n_entries):
        order = np.ascontiguousarray(seq_of_block, np.uint32)
        check(_set_bp_dev(self._h, len(order), order.ctypes.data, rowptr, col, val, n_entries))

    def stage_timing(self, enable=True):
        """per-kernel device timings on / off (dafs_hip_stage_timing: HIP events around every launch of the library)"""
        check(_stage_timing(self._h, 1 if enable else 0))

    def stage_report(self):
        """{kernel: (ms summed over its launches, longest launch ms, launches)} since the last report"""
        buf = (StageTime * 32)()
        n = C.c_uint32()
        check(_stage_report(self._h, buf, 32, C.byref(n)))
        return {buf[k].kernel.decode(): (buf[k].ms, buf[k].longest_ms, buf[k].launches) for k in range(min(n.value, 32))}

    def nodes_demotions(self):
        """split-mode nodes that lost their folding workgroups and went on in the one-workgroup form (since nodes_close)"""
        n = C.c_uint32()
        check(_nodes_demotions(self._h, C.byref(n)))
        return n.value

    def update_basepairing(self, seq, mask, ss):
        """DAFS::update_basepairing_probability (--bp-update): the L x L matrix re-estimated under the structure ss"""
        seq = np.ascontiguousarray(seq, np.uint32); mask = np.ascontiguousarray(mask, np.uint8)
        ss = np.ascontiguousarray(ss, np.uint32)
        n, L = mask.shape
        p = np.zeros((L, L), np.float32)
        check(_update_basepairing(self._h, n, L, seq.ctypes.data, mask.ctypes.data, ss.ctypes.data, p.ctypes.data))
        return p

    def consensus_structure(self, seq, mask, th, want_p=False):
        seq = np.ascontiguousarray(seq, np.uint32); mask = np.ascontiguousarray(mask, np.uint8)
        n, L = mask.shape
        ss = np.zeros(L, np.uint32)
        score = C.c_float()
        p = np.zeros((L, L), np.float32) if want_p else None
        check(_consensus_structure(self._h, n, L, seq.ctypes.data, mask.ctypes.data, th, ss.ctypes.data, C.byref(score),
                                   None if p is None else p.ctypes.data))
        return np.float32(score.value), ss, p


def build_tree(sim):
    """DAFS::build_tree (host code in the library): (score, left, right) with -1 for leaves"""
    sim = np.ascontiguousarray(sim, np.float32)
    n = sim.shape[0]
    score = np.zeros(2 * n - 1, np.float32); left = np.zeros(2 * n - 1, np.int32); right = np.zeros(2 * n - 1, np.int32)
    check(_build_tree(n, sim.ctypes.data, score.ctypes.data, left.ctypes.data, right.ctypes.data))
    return score, left.astype(np.int64), right.astype(np.int64)


def dd_params(**kw):
    p = DDParams()
    _dd_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def make_brackets(ss):
    ss = np.ascontiguousarray(ss, np.uint32)
    buf = C.create_string_buffer(len(ss) + 1)
    _make_brackets(len(ss), ss.ctypes.data, buf)
    return buf.value.decode()
