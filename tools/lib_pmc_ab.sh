#!/bin/bash
# A/B of library builds with counters: LIBS="build/lib_a.so build/lib_b.so" tools/lib_pmc_ab.sh -> kernel ms, traffic, issue per shape
for cfg in "262144 rk4 0" "65536 rk45 1" "65536 rk4 1" "1048576 rk4 0" "4096 rk45 0"; do
  set -- $cfg
  for lib in $LIBS; do
    STG_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --also 0 --envs-per-gpu $1 --solver $2 --thermal $3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('$(basename $lib) envs $1 $2 thermal $3: kernel %.4f ms  step %.4f ms  traffic %s MB (alg %.1f)  issue %s' % (r['kernel_ms_avg'], b['ms_per_step'], None if r['traffic'] is None else round(r['traffic']/1e6,1), r['algorithmic_bytes']/1e6, r['valu_issue_frac']))"
  done
done
