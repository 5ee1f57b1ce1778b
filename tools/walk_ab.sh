#!/bin/bash
# A/B of the sorted schedule's tile walk (STG_WALK_TILES = tiles an XCD group keeps in flight): kernel time and HBM traffic
# (live PMC passes of bench.py) per launch size.  usage (gpurun): bash tools/walk_ab.sh <out-file>
out=${1:-gpurun_out/walk_ab.txt}; mkdir -p $(dirname $out); : > $out
for cfg in "262144 rk4 0" "1048576 rk4 0" "262144 rk45 1" "131072 rk45 1"; do
  set -- $cfg
  for w in 1 2 4 8 64; do
    STG_WALK_TILES=$w timeout -k 10 200 python3 bench.py --steps 6 --warmup 1 --cpu-baseline 0 --also 0 --envs-per-gpu $1 --solver $2 --thermal $3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('envs $1 $2 thermal $3 walk $w: kernel %.4f ms  traffic %s MB (algorithmic %.1f MB)  issue %s' % (r['kernel_ms_avg'], None if r['traffic'] is None else round(r['traffic']/1e6,1), r['algorithmic_bytes']/1e6, r['valu_issue_frac']))" >> $out
  done
done
cat $out
