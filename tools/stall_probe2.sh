#!/bin/bash
# Round 3: does the late fence return coincide with CPU-bandwidth throttling of the container (cgroup cpu.stat)?
set -o pipefail
OUT=${1:-gpurun_out/r03c}
mkdir -p "$OUT"
python3 tools/sync_wait_probe.py --blocks 300 --cpu-burst 4000000 > $OUT/probe_burst.txt 2>&1 || { tail -20 $OUT/probe_burst.txt; exit 1; }
echo "burst probe done"
python3 tools/sync_wait_probe.py --blocks 300 --cpu-burst 4000000 --threads 4 > $OUT/probe_burst_4threads.txt 2>&1 || { tail -20 $OUT/probe_burst_4threads.txt; exit 1; }
echo "burst probe, 4 threads, done"
