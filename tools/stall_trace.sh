#!/bin/bash
# Round 3: find the ~80 ms host-side stall bench.py used to re-time around.
# 1) plain run with STG_BENCH_DEBUG=1 (per-block wall vs device span, per-step host stamps, every Python GC pass);
# 2) the same command under rocprofv3 --hip-trace --kernel-trace (which HIP call returns late, if any).
set -o pipefail
OUT=${1:-gpurun_out/r03a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
STG_BENCH_DEBUG=1 python3 $R/bench.py --cpu-baseline 0 --pmc off > $R/$OUT/debug_plain.json 2> $R/$OUT/debug_plain.err || exit 1
echo "plain run done"
STG_BENCH_DEBUG=1 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --cpu-baseline 0 --pmc off > $R/$OUT/debug_traced.json 2> $R/$OUT/debug_traced.err || exit 1
echo "traced run done"
ls -la $R/$OUT/trace/*/ | head -20
