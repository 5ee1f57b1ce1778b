#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (csrc/resource_usage.txt) as one line per kernel."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "spin-torque-rl-gym_amd/csrc/resource_usage.txt"
cur = None
rows = []
for line in open(path):
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and key not in cur:
            cur[key] = int(m.group(1))
for r in rows:
    print(f"{r['name'][:70]:70s} vgpr={r.get('VGPRs')} sgpr={r.get('SGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} "
          f"occ={r.get('Occupancy [waves/SIMD]')} sspill={r.get('SGPRs Spill')} vspill={r.get('VGPRs Spill')} lds={r.get('LDS Size [bytes/block]')}")
