#!/bin/bash
# Hybrid wave-specialised launch (65 536 < N <= 131 072 envs, thermal): on (default) against off (STG_HYBRID=0), kernel ms.
# bash tools/hybrid_ab.sh [reps] [solver]
reps=${1:-2}; solver=${2:-rk45}
for rep in $(seq $reps); do
 for mode in on off; do
  for n in 65600 66000 69632 73728 81920 90112 98304 106496 114688 122880 126976 131072; do
   if [ $mode = off ]; then export STG_HYBRID=0; else unset STG_HYBRID; fi
   python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --also 0 --pmc off --solver $solver --thermal 1 --envs-per-gpu $n 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('hybrid $mode $solver n=$n: kernel %.4f ms' % d['roofline']['kernel_ms_avg'])"
  done
 done
done
