#!/bin/bash
# Widened randomised HIP-vs-oracle sweep (tests/test_gpu_parity.py::test_randomised_configurations_vs_oracle with other parameter draws):
# bash tools/random_sweep.sh <first offset> <last offset> [cases]  -> gpurun_out/random_sweep.txt (one line per case and step)
first=${1:-1}; last=${2:-8}; cases=${3:-6}
out=gpurun_out/random_sweep.txt; mkdir -p gpurun_out; : > $out
for o in $(seq $first $last); do
  for s in rk4 euler rk45; do
    echo "== solver $s offset $o" >> $out
    timeout -k 10 300 python3 tools/diag_random_case.py $s $o $cases >> $out 2>&1 || { echo "FAILED $s $o" >> $out; exit 1; }
  done
done
python3 - <<'PY'
import re
worst = {}
for ln in open("gpurun_out/random_sweep.txt"):
    m = re.match(r"case (\d+) thermal=(\w+) step \d: worst \|dm\| hip-oracle ([0-9.e+-]+|nan).*wave_spec on-off ([0-9.e+-]+)", ln)
    if ln.startswith("=="): key = ln.split()[2]
    if m:
        k = (key, m.group(2)); worst[k] = max(worst.get(k, 0.0), float(m.group(3)))
        if float(m.group(4)) != 0.0: print("wave_spec on/off differ:", ln.strip())
for k in sorted(worst): print("worst |dm| hip-oracle  solver %-5s thermal=%-5s : %.3e" % (k[0], k[1], worst[k]))
PY
