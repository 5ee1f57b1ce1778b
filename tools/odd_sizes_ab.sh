#!/bin/bash
# Batch sizes that are no multiple of 8 tiles (32768 envs) / of a tile (4096): LIBS="build/lib_prev.so build/lib_new.so" tools/odd_sizes_ab.sh
for lib in $LIBS; do
  SOLVER=rk45 LIBS=$lib bash tools/ab_sizes.sh 1 "69632 1" "81920 1" "100000 1" "114688 1" "131072 1" "200000 1" "300000 1" "200000 0" "100000 0" "81920 0"
  SOLVER=rk4 LIBS=$lib bash tools/ab_sizes.sh 1 "81920 0" "100000 0" "200000 0" "300000 0" "100000 1" "200000 1" "262144 0" "65536 1"
done
