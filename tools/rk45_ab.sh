#!/bin/bash
# A/B of library builds on the RK45 shapes (alternating, inside ONE gpurun call): LIBS="build/lib_a.so build/lib_b.so" tools/rk45_ab.sh [reps]
reps=${1:-2}
for rep in $(seq $reps); do
for lib in $LIBS; do
 for cfg in "65536 1" "4096 0" "262144 1"; do
  set -- $cfg
  STG_HIP_LIBRARY=$PWD/$lib python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --also 0 --pmc off --solver rk45 --thermal $2 --envs-per-gpu $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $lib) rk45 n=$1 thermal=$2: kernel %.4f ms' % d['roofline']['kernel_ms_avg'])"
 done
done
done
