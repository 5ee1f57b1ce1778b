#!/bin/bash
# usage (through gpurun): bash tools/cheapn_ab.sh > gpurun_out/cheapn_ab.txt
for rep in 1 2; do
  python3 tools/cheapn_ab.py || exit 1
  STG_HIP_LIBRARY=$PWD/build/lib_cheapn.so python3 tools/cheapn_ab.py || exit 1
done
