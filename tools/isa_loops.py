#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing (tuning aid).

  python tools/isa_loops.py file.s [kernel-name-substring]
For every innermost loop (a backward branch to a label): instruction count by class and the vector-pipe cycles
they need on gfx950 (2 per wave64 instruction, 4 for packed-f32 and f64 arithmetic: MI355X_MICROARCH.md).
"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
sel = sys.argv[2] if len(sys.argv) > 2 else None
if sel:
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sel in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section"))
    lines = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(lines):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
# innermost loops, and every loop of 100+ lines (a sweep whose body holds a rare inner branch-back still shows up)
inner = [a for a in loops if a[1] - a[0] >= 100 or not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in loops)]
for lo, hi in sorted(set(inner)):
    ops = [l.split()[0] for l in lines[lo:hi + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    c = Counter(ops)
    valu = {k: v for k, v in c.items() if k.startswith("v_")}
    slow = sum(v for k, v in valu.items() if k.startswith("v_pk_") or "_f64" in k)
    n_valu = sum(valu.values())
    cyc = 2 * (n_valu - slow) + 4 * slow
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
    salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop")))
    print("loop lines %d-%d: %d instr | valu %d (packed/f64 %d) -> %d pipe cycles | lds %d | vmem %d | salu %d | waitcnt %d | nop %d" %
          (lo, hi, len(ops), n_valu, slow, cyc, lds, vmem, salu, c.get("s_waitcnt", 0), c.get("s_nop", 0)))
    print("   ", ", ".join("%s %d" % kv for kv in sorted(valu.items(), key=lambda kv: -kv[1])[:14]))
