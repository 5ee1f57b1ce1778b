#!/bin/bash
# A/B of the alternating pair placement for wave-specialised launches with three or four workgroups per CU (32 768 < N < 65 536 envs):
# STG_SNAKE=0 switches it off (= plain rank-major order, the behaviour before).  bash tools/pair_ab.sh [reps] -> stdout
reps=${1:-3}
for rep in $(seq $reps); do
 for mode in new old; do
  for cfg in "rk45 36864" "rk45 45000" "rk45 50000" "rk45 60000" "rk45 65536" "rk4 50000" "rk4 60000" "euler 60000"; do
   set -- $cfg
   if [ $mode = old ]; then export STG_SNAKE=0; else unset STG_SNAKE; fi
   python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --also 0 --pmc off --solver $1 --thermal 1 --envs-per-gpu $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$mode $1 n=$2 thermal: kernel %.4f ms' % d['roofline']['kernel_ms_avg'])"
  done
 done
done
