#!/bin/bash
# kernel ms over launch sizes for one (solver, thermal), wave_spec auto/on/off: tools/size_sweep.sh rk45 1 "4096 32768 65536 131072"
solver=$1; th=$2; sizes=${3:-"4096 16384 32768 65536 131072 262144"}
for n in $sizes; do
 for ws in auto on off; do
  python3 bench.py --steps 4 --warmup 1 --cpu-baseline 0 --also 0 --solver $solver --thermal $th --envs-per-gpu $n --wave-spec $ws 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$solver thermal=$th n=$n wave_spec=$ws: %.3e env-steps/s  kernel %.3f ms' % (d['value'], r['kernel_ms_avg']))"
 done
done
