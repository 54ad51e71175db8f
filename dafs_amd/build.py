"""Build libdafs_hip.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc.

Usage: python -m dafs_amd.build [--force]
The .so is written next to the sources (dafs_amd/libdafs_hip.so); it is git-ignored but travels
to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdafs_hip.so")
OBJ = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference CPU build has no FMA contraction; bit-exact parity needs the same.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-Wno-missing-braces", "-fno-fast-math"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")) and not f.startswith("cli_"))


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    mt = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [src, os.path.join(HERE, "..", "include", "dafs_hip.h")]
    return any(os.path.getmtime(d) > mt for d in deps)


def _compile(name, force):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name + ".o")
    if not force and not _stale(src, obj):
        return obj
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if name.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (name, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
