#!/usr/bin/env python3
"""Registers, spills, scratch and the occupancy the code object allows, per kernel of one .hip file.

  python tools/kernel_resources.py dafs_amd/csrc/pairhmm3.hip [more.hip ...] [-- extra hipcc flags]

Compiles the device side only (gfx950, the flags of dafs_amd/build.py) with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel.  Runs without a GPU.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dafs_amd import build as b  # noqa: E402


def demangle(names):
    r = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    out = r.stdout.strip().split("\n") if r.returncode == 0 else names
    return [re.sub(r"\(.*$", "", re.sub(r"^void ", "", o)).replace("dafs::", "") for o in out]


def resources(src, extra=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = [b.HIPCC] + b.FLAGS + b.FILE_FLAGS.get(os.path.basename(src), []) + list(extra) + ["--cuda-device-only", "-c", src, "-o", os.path.join(td, "x.co"),
                                                    "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(r.stderr)
    rows, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark: [^ ]* *(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        extra = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    print("%-44s %5s %5s %7s %7s %8s %6s %5s" % ("kernel", "vgpr", "agpr", "sgprspl", "vgprspl", "scratchB", "ldsB", "w/EU"))
    for src in args:
        rows = resources(src, extra)
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            print("%-44s %5s %5s %7s %7s %8s %6s %5s" % (n[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                      r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                                      r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
