#!/bin/bash
# A/B of library builds: tools/lib_ab.sh <lib.so>... ; prints kernel ms for the thermal headline shapes, wave_spec off/on
for lib in "$@"; do
 for cfg in rk4 rk45; do
  for ws in off on; do
   STG_HIP_LIBRARY=$lib python3 bench.py --steps 4 --warmup 1 --cpu-baseline 0 --also 0 --solver $cfg --thermal 1 --wave-spec $ws 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$(basename $lib) $cfg wave_spec=$ws: kernel %.3f ms' % r['kernel_ms_avg'])"
  done
 done
done
