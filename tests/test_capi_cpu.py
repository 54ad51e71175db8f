"""CPU tests of the C-ABI library: it loads, exports every symbol include/dafs_hip.h declares,
and its host-only helpers (plan, residue codes, model tables) behave.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dafs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dafs_hipk?_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dafs_amd import capi
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(capi.lib, n), "libdafs_hip.so does not export %s" % n


def test_strerror_and_codes():
    from dafs_amd import capi
    assert capi._strerror(0) == b"ok"
    assert b"overflow" in capi._strerror(-5)
    assert b"unknown" in capi._strerror(-99)


def test_pair_ranges_of_the_sharded_phase_agree_with_the_python_driver():
    """dafs_hip_pair_range (what dafs --devices shards by) = dist.pair_ranges (what the torch driver shards by): contiguous,
    covering, in rank order -- so the ranks' shards concatenated in rank order are the whole in pair order"""
    from dafs_amd import capi, dist
    assert capi._strerror(-7).startswith(b"dafs_hip: the collective")
    f = capi.lib.dafs_hip_pair_range
    f.restype = None
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    for npairs in (0, 1, 3, 8128, 130816, 10 ** 12 + 7):
        for world in (1, 2, 3, 8, 64):
            want = dist.pair_ranges(npairs, world)
            for r in range(world):
                b, e = C.c_uint64(), C.c_uint64()
                f(npairs, world, r, C.byref(b), C.byref(e))
                assert (b.value, e.value) == (want[r], want[r + 1]), (npairs, world, r)


def test_residue_codes():
    from dafs_amd import capi
    assert list(capi.encode("ACGUTNacgutn")) == [0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5]
    assert list(capi.encode("X-~@z")) == [6] * 5


def test_default_model_matches_oracle_tables(oracle):
    from dafs_amd import capi
    m = capi.Pairhmm3Model()
    capi.pairhmm3_default_model(C.byref(m))
    init = np.zeros(3, np.float32); trans = np.zeros(9, np.float32)
    match = np.zeros(256 * 256, np.float32); ins = np.zeros(256, np.float32)
    oracle.lib.orc_probcons_tables.argtypes = [C.c_void_p] * 4
    oracle.lib.orc_probcons_tables(init.ctypes.data, trans.ctypes.data, match.ctypes.data, ins.ctypes.data)
    match = match.reshape(256, 256)
    assert np.array_equal(np.array(list(m.init), np.float32), init)
    t = np.array([list(r) for r in m.trans], np.float32)
    finite = np.isfinite(trans.reshape(3, 3))
    assert np.array_equal(t[finite], trans.reshape(3, 3)[finite])
    for a in range(256):
        ca = capi.residue_code(bytes([a]))
        assert np.float32(m.ins[ca]) == ins[a]
        for b in range(0, 256, 3):
            cb = capi.residue_code(bytes([b]))
            assert np.float32(m.match[ca][cb]) == match[a][b], (a, b)


def test_plan_covers_lengths():
    from dafs_amd import capi
    p = capi.PairhmmPlan()
    for ntasks, l1, l2 in ((1, 1, 1), (45, 82, 82), (496, 80, 80), (8128, 160, 160), (32640, 214, 214), (130816, 428, 428), (3, 2000, 2047)):
        assert capi.pairhmm_plan(ntasks, l1, l2, p) == 0
        assert p.group in (16, 32, 64) and p.group * p.width >= l2 + 1
        assert p.slab_steps == l1 + p.group and p.nwaves % 4 == 0 and p.nwaves <= 8 * 1024
        assert p.scratch_bytes == p.nwaves * p.slab_steps * p.width * 64 * 4 * 2  # DP slab + entry lists
    assert capi.pairhmm_plan(1, 10, 5000, p) == -4
    assert capi.pairhmm_plan(0, 10, 10, p) == -1


def test_host_build_tree_matches_python_twin():
    """dafs_host_build_tree (host code inside the library, used by the CLI and the pipeline driver) against the
    Python restatement of DAFS::build_tree, on random similarity matrices with ties."""
    from dafs_amd import capi, pipeline
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 9, 33):
        s = rng.integers(0, 50, size=(n, n)).astype(np.float32) / np.float32(64)
        s = np.maximum(s, s.T)
        np.fill_diagonal(s, 1.0)
        a = capi.build_tree(s)
        b = pipeline.build_tree(s)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_driver_refuses_option_combinations_it_does_not_implement():
    """pipeline.run: level batches with --bp-update, or a sharded phase 1 with supplied posteriors / the four-way transform,
    raise instead of silently dropping the option (no GPU is touched before the check)"""
    import pytest
    from dafs_amd import pipeline
    with pytest.raises(ValueError):
        pipeline.run(["a", "b"], ["ACGU", "ACGU"], ctx=object(), level_sync=True, bp_update=True)
    for kw in (dict(bp=[]), dict(mp=()), dict(w_pct_f=0.3)):
        with pytest.raises(ValueError):
            pipeline.run(["a", "b"], ["ACGU", "ACGU"], ctx=object(), shard=(None, None), **kw)
