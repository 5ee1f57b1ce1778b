#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py's headline kernel, one pass per counter group (the MI355X guide:
# FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains besides --kernel-trace next to --pmc).
# usage: tools/pmc_passes.sh <out-dir> [bench args...]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "FETCH_SIZE" "WRITE_SIZE" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/pass$i" -- python3 bench.py --cpu-baseline 0 --also 0 "$@" > "$out/pass$i.json" 2> "$out/pass$i.err" || echo "pass $i ($pass) failed: $(tail -2 $out/pass$i.err)"
done
find "$out" -name "*counter_collection.csv" | head
