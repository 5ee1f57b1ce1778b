#!/bin/bash
# A/B of library builds on the RK45 shapes: LIBS="a.so b.so" tools/rk45_ab.sh
for lib in $LIBS; do
 for cfg in "65536 1" "65536 0" "4096 0"; do
  set -- $cfg
  STG_HIP_LIBRARY=$PWD/$lib python3 bench.py --steps 6 --warmup 1 --cpu-baseline 0 --also 0 --solver rk45 --thermal $2 --envs-per-gpu $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $lib) rk45 n=$1 thermal=$2: kernel %.3f ms' % d['roofline']['kernel_ms_avg'])"
 done
done
