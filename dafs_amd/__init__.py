"""dafs_amd -- MI355X (gfx950) implementation of the DAFS probability-matrix and
dual-decomposition hot path, behind a C ABI (include/dafs_hip.h).

This package is only the thin Python side used by tests and bench.py: a ctypes binding to
libdafs_hip.so.  There is no CPU fallback; importing `dafs_amd.capi` raises if the library has
not been built (python -m dafs_amd.build).
"""
__all__ = ["capi", "build", "synth"]
