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
FLAGS += os.environ.get("DAFS_HIP_EXTRA_FLAGS", "").split()  # tuning experiments only
# Per-file flags.  The pair kernels run four or more wavefronts per SIMD, where a packed-f32 instruction costs as much
# as two plain ones and plain f32 adds/multiplies issue 2.5x faster than anything else (profiles/r02_a_valu_rate.txt):
# the SLP vectoriser's automatic v_pk_* pairs (plus the moves that line their operands up) are a loss there.
FILE_FLAGS = {"pairhmm3.hip": ["-fno-slp-vectorize"], "pairhmm5.hip": ["-fno-slp-vectorize"]}


def source_sha16(names=("pairhmm3.hip", "pairhmm5.hip", "pair_sweeps.h", "pc_math.h", "contra_math.h")):
    """sha256 (first 16 hex digits) of the pair kernels' sources: ties a committed PMC summary to the build it measured"""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


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
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(name, []) + (["-x", "hip"] if name.endswith(".cpp") else []) + ["-c", src, "-o", obj]
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


HOST = os.path.join(CSRC, "host")
CLI = os.path.join(HERE, "dafs")
SELFTEST = os.path.join(HERE, "plugin_selftest")


def build_cli(force=False):
    """The C++ host side: the `dafs` command line and the plugin self-test, linked against libdafs_hip.so."""
    cxx = os.environ.get("CXX", "g++")
    common = [os.path.join(HOST, "fasta.cpp")]
    targets = ((CLI, common + [os.path.join(HOST, "cli_main.cpp")]),
               (SELFTEST, common + [os.path.join(HOST, "plugins.cpp"), os.path.join(HOST, "plugin_selftest.cpp")]))
    deps = [os.path.join(HOST, f) for f in os.listdir(HOST)] + [os.path.join(HERE, "..", "include", "dafs_hip.h"), OUT]
    for out, srcs in targets:
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
            continue
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd = [cxx, "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(rocm, "include"), "-o", out] + srcs
        cmd += ["-L" + HERE, "-ldafs_hip", "-Wl,-rpath,$ORIGIN"]
        if out == CLI:  # the staging copies of --devices; librccl itself is loaded on demand
            cmd += ["-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-ldl", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host build failed:\n" + r.stderr)
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_cli(force="--force" in sys.argv))
