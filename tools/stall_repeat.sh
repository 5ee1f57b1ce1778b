#!/bin/bash
# Round 3: run the benchmark's rows several times with STG_BENCH_DEBUG=1; every timed block prints its wall time, the device's
# own span and the container's cgroup cpu.stat delta (nr_throttled / throttled_usec) over the block.
set -o pipefail
OUT=${1:-gpurun_out/r03d}; N=${2:-6}; shift $(( $# < 2 ? $# : 2 ))   # the rest: NAME=value knobs for the run, e.g. STG_SNAKE=0
mkdir -p "$OUT"
for i in $(seq 1 $N); do
  env STG_BENCH_DEBUG=1 "$@" python3 bench.py --cpu-baseline 0 --pmc off > $OUT/run$i.json 2> $OUT/run$i.err || { tail -5 $OUT/run$i.err; exit 1; }
  echo "run $i done"
done
