#!/bin/bash
# A/B of the wave-specialised thermal kernels: kernel ms at several launch sizes, wave_spec off vs on
for n in 4096 65536 131072 262144; do
 for cfg in "rk4" "rk45"; do
  for ws in off on; do
  python3 bench.py --steps 4 --warmup 1 --cpu-baseline 0 --also 0 --solver $cfg --thermal 1 --envs-per-gpu $n --wave-spec $ws 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('$cfg n=$n wave_spec=$ws: %.3e env-steps/s  kernel %.3f ms  frac %.4f' % (d['value'], r['kernel_ms_avg'], r['frac']))"
  done
 done
done
