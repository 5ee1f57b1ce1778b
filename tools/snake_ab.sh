#!/bin/bash
# A/B of the boustrophedon workgroup order (STG_SNAKE: 0 off, 1 on, unset = automatic): usage tools/snake_ab.sh "<envs solver thermal>" ...
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2; do for sn in 0 1 auto; do
    if [ $sn = auto ]; then unset STG_SNAKE; else export STG_SNAKE=$sn; fi
    timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --also 0 --pmc off --envs-per-gpu $1 --solver $2 --thermal $3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('snake $sn envs $1 $2 thermal $3: kernel %.4f ms' % r['kernel_ms_avg'])"
  done; done
done
