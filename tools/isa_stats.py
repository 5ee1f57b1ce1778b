#!/usr/bin/env python3
"""Instruction mix of one kernel in the hipcc -S output: python tools/isa_stats.py <file.s> <kernel-substring>"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and ":" in l and key in l.split(":")[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
ops = collections.Counter()
for l in body:
    l = l.strip()
    if not l or l.startswith(".") or l.startswith(";") or l.endswith(":"):
        continue
    ops[l.split()[0]] += 1
total = sum(ops.values())
print(f"{lines[start]} total instructions: {total}")
groups = collections.Counter()
for op, c in ops.items():
    if re.match(r"v_(fma|mul|add|fmac)_f64", op):
        groups["v_f64 arith"] += c
    elif re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|frexp|ldexp|trig|fract|floor|ceil|rndne|cmp.*f64|cndmask)", op) and "f64" in op:
        groups["v_f64 special"] += c
    elif op.startswith("v_") and "f32" in op:
        groups["v_f32"] += c
    elif op.startswith("v_readlane") or op.startswith("v_writelane"):
        groups["v_read/writelane (sgpr spill)"] += c
    elif op.startswith("v_"):
        groups["v_other"] += c
    elif op.startswith("s_"):
        groups["scalar"] += c
    elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        groups["vmem"] += c
    elif op.startswith("ds_"):
        groups["lds"] += c
    else:
        groups["other"] += c
for g, c in groups.most_common():
    print(f"  {g:32s} {c}")
print("  top ops:", ", ".join(f"{o}={c}" for o, c in ops.most_common(25)))
