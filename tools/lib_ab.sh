#!/bin/bash
# A/B of library builds: tools/lib_ab.sh <lib.so>... ; prints kernel ms for the headline shapes
for lib in "$@"; do
 for cfg in "rk4 0" "rk4 1" "rk45 0" "rk45 1"; do
  set -- $cfg
  STG_HIP_LIBRARY=$lib python3 bench.py --steps 4 --warmup 1 --cpu-baseline 0 --also 0 --solver $1 --thermal $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$(basename $lib) $1 thermal=$2: kernel %.3f ms' % r['kernel_ms_avg'])"
 done
done
