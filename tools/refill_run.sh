#!/bin/bash
# lane refill: correctness (bit-identity with the one-env-per-lane launch) and timings of STG_REFILL=<envs per lane>,<check>
set -o pipefail
timeout -k 10 300 python3 tools/refill_check.py 2,16 131072 70001 9000 || exit 1
timeout -k 10 300 python3 tools/refill_check.py 3,64 200001 || exit 1
for e in "STG_REFILL=0" "STG_REFILL=2,64" "STG_REFILL=3,64" "STG_REFILL=4,64"; do
  ENVV="$e" LIBS="${LIB:-build/lib_new.so}" bash tools/ab_sizes.sh 1 "131072 1" "196608 1" "262144 1"
done
