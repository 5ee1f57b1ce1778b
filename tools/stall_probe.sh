#!/bin/bash
# Round 3: blocked-wait vs spin-wait fence statistics (tools/sync_wait_probe.py), plain and with hipDeviceScheduleSpin
set -o pipefail
OUT=${1:-gpurun_out/r03b}
mkdir -p "$OUT"
python3 tools/sync_wait_probe.py --blocks 1500 > $OUT/sync_wait_probe.txt 2>&1 || { tail -20 $OUT/sync_wait_probe.txt; exit 1; }
echo "probe 1 done"
python3 tools/sync_wait_probe.py --blocks 1500 --spin-flag > $OUT/sync_wait_probe_spinflag.txt 2>&1 || { tail -20 $OUT/sync_wait_probe_spinflag.txt; exit 1; }
echo "probe 2 done"
python3 tools/sync_wait_probe.py --blocks 400 --idle-ms 5 > $OUT/sync_wait_probe_idle5ms.txt 2>&1 || { tail -20 $OUT/sync_wait_probe_idle5ms.txt; exit 1; }
echo "probe 3 done"
