#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter over its dispatches.
usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "stg_step_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if key not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    print(name)
    for c, vals in sorted(ctrs.items()):
        print(f"   {c:28s} mean={sum(vals)/len(vals):.6g}  n={len(vals)}")
