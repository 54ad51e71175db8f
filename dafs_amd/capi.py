"""ctypes binding of include/dafs_hip.h (the same entry points a cgo/JNI/C++ shim would bind)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdafs_hip.so")

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
_consistency = _sig("dafs_hip_consistency", C.c_int, [C.c_void_p, C.c_float, C.c_float])
pairhmm_plan = _sig("dafs_hipk_pairhmm_plan", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PairhmmPlan)])
pairhmm3_launch = _sig("dafs_hipk_pairhmm3_launch", C.c_int, [C.POINTER(Pairhmm3Args), C.POINTER(PairhmmPlan), C.c_void_p])
pairhmm3_default_model = _sig("dafs_hip_pairhmm3_default_model", None, [C.POINTER(Pairhmm3Model)])
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

    def consistency(self, w_pct_a=0.25, w_pct_s=0.25):
        check(_consistency(self._h, w_pct_a, w_pct_s))
