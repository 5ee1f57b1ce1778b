#!/bin/bash
# A/B of library builds on the RK45 shapes, alternating inside ONE gpurun call (boxes differ by up to 8 %):
#   LIBS="build/lib_a.so build/lib_b.so" [ENVV="STG_REFILL=2"] tools/ab_sizes.sh [reps] ["n thermal" ...]
reps=${1:-2}; shift
cfgs=("$@"); [ ${#cfgs[@]} -eq 0 ] && cfgs=("4096 0" "65536 1" "131072 1" "262144 1")
for rep in $(seq $reps); do
for lib in $LIBS; do
 for cfg in "${cfgs[@]}"; do
  set -- $cfg
  env $ENVV STG_HIP_LIBRARY=$PWD/$lib python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --also 0 --pmc off --solver ${SOLVER:-rk45} --thermal $2 --envs-per-gpu $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $lib) $ENVV ${SOLVER:-rk45} n=$1 thermal=$2: kernel %.4f ms  (wall/step %.4f)' % (d['roofline']['kernel_ms_avg'], d['ms_per_step']))"
 done
done
done
