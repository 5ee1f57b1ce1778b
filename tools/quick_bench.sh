#!/bin/bash
# quick A/B numbers: kernel ms and env-steps/s for the four (solver, thermal) combinations at 65536 envs
for cfg in "rk4 0" "rk4 1" "rk45 0" "rk45 1"; do
  set -- $cfg
  python3 bench.py --steps 4 --warmup 1 --cpu-baseline 0 --also 0 --solver $1 --thermal $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('$1 thermal=$2: %.3e env-steps/s  kernel %.3f ms  frac %.4f  work/step %.1f' % (d['value'], r['kernel_ms_avg'], r['frac'], r['work_units_per_env_step']))"
done
