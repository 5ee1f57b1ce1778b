#!/bin/bash
# Collects one round's evidence on a GPU box: rocprofv3 --kernel-trace --stats of bench.py, the PMC passes, a plain bench
# line, and the probe outputs DESIGN.md quotes.  usage (through gpurun): bash tools/collect_profiles.sh <tag>
set -u
tag=${1:-rXX}
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-baseline 0 > $out/bench_under_rocprof.json 2> $out/stats.err || echo "stats run failed"
bash tools/pmc_passes.sh $out/pmc --steps 3 --warmup 1 > $out/pmc.log 2>&1
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || echo "bench failed"
{
  echo "== tools/probes/clock_vs_load (shader clock and fp64 FMA issue vs number of wavefronts)"
  timeout -k 10 60 tools/probes/clock_vs_load
  echo "== tools/probes/clock_vs_load 200000 3 (sustained)"
  timeout -k 10 60 tools/probes/clock_vs_load 200000 3
  if [ -f build/lib_prof.so ]; then
    echo "== tools/probe_wave_records.py 65536 1 3 8 (library built with -DSTG_PROFILE_LOOP)"
    STG_HIP_LIBRARY=$PWD/build/lib_prof.so timeout -k 10 120 python3 tools/probe_wave_records.py 65536 1 3 8
    echo "== tools/probe_wave_records.py 65536 0 2 8"
    STG_HIP_LIBRARY=$PWD/build/lib_prof.so timeout -k 10 120 python3 tools/probe_wave_records.py 65536 0 2 8
  fi
} > $out/probes.txt 2>&1
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || echo "smoke failed"
tail -2 $out/smoke.txt; cat $out/pmc_summary.txt | head -20; head -3 $out/stats/*/*kernel_stats.csv
