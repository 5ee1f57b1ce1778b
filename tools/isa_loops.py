#!/usr/bin/env python3
"""Find loops (backward branches) in one kernel of the hipcc -S output and print each loop's instruction mix.
usage: python tools/isa_loops.py <file.s> <kernel-substring> [min_len]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))       # (a kernel may hold several s_endpgm)
body = lines[start:end + 1]
labels, insts = {}, []
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if t.endswith(":"):
        continue
    insts.append(t)


def mix(seq):
    g = collections.Counter()
    for t in seq:
        op = t.split()[0]
        if re.match(r"v_(fma|mul|add|fmac)_f64", op): g["f64 arith"] += 1
        elif "f64" in op: g["f64 other(" + op.split("_f64")[0][2:] + ")"] += 1
        elif op.startswith("v_") and ("f32" in op): g["f32(" + op[2:].split("_f32")[0] + ")"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_writelane"): g["read/writelane"] += 1
        elif op.startswith("v_cndmask"): g["cndmask"] += 1
        elif op.startswith("v_"): g["v_int/other"] += 1
        elif op.startswith("s_"): g["scalar"] += 1
        else: g["mem/other"] += 1
    return g


loops = []
for idx, t in enumerate(insts):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt <= idx and idx - tgt >= min_len:
            loops.append((tgt, idx))
print(f"{len(insts)} instructions, {len(loops)} loops >= {min_len}")
for a, b in loops:
    g = mix(insts[a:b + 1])
    valu = sum(c for k, c in g.items() if k not in ("scalar", "mem/other"))
    print(f"loop [{a},{b}] len={b - a + 1} VALU={valu}: " + ", ".join(f"{k}={c}" for k, c in g.most_common()))
