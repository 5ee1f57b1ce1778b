#!/bin/bash
# Lane-refill sweep (round 3): kernel ms per launch for envs-per-lane R and attempts-between-refill-points, one library
LIB=${LIB:-build/lib_new.so}
for e in "STG_REFILL=0" "STG_REFILL=4,16" "STG_REFILL=4,64" "STG_REFILL=8,16" "STG_REFILL=8,64" "STG_REFILL=16,32"; do
  ENVV="$e" LIBS="$LIB" bash tools/ab_sizes.sh 1 "262144 1" "524288 1" "1048576 1" "1048576 0"
done
