#!/usr/bin/env python3
"""List the loops (backward branches) of one kernel in hipcc -S output with instruction-class counts.
usage: isa_loops.py file.s kernel_symbol"""
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
labels, insts = {}, []
for i in range(start, end):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not l or l.startswith((";", ".", "//")):
        continue
    insts.append(l.split(";")[0].strip())


def cls(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "mem"
    return "other"


loops = []
for idx, ins in enumerate(insts):
    m = re.match(r"^s_c?branch\S*\s+(\.LBB[0-9_]+)", ins)
    if m and m.group(1) in labels and labels[m.group(1)] <= idx:
        loops.append((labels[m.group(1)], idx, m.group(1)))
print(f"{sym}: {len(insts)} instructions, {len(loops)} loops")
for a, b, lab in sorted(loops):
    c = {}
    for ins in insts[a:b + 1]:
        k = cls(ins.split()[0])
        c[k] = c.get(k, 0) + 1
    print(f"  {lab:14s} [{a:6d},{b:6d}] n={b - a + 1:6d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
