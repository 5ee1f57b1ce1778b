#!/bin/bash
# One round's evidence on a GPU box (usage, through gpurun: bash tools/collect_round.sh <tag> [tests]):
#   1. bench.py as the driver runs it -- its own live PMC passes included -- with the per-row counter table dumped
#   2. rocprofv3 --kernel-trace --stats of the same command (counters off): per-kernel durations
#   3. a two-rank rehearsal of the multi-GPU path on the one GPU (gloo; the RCCL 8-GPU run is the driver's), self-launched and under torch.distributed.run
#   4. smoke()
# Results under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
tag=${1:-rXX}
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${2:-}" = "tests" ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=15 > $out/pytest_gpu.log 2>&1 || { echo "GPU TESTS FAILED"; tail -30 $out/pytest_gpu.log; exit 1; }
  tail -3 $out/pytest_gpu.log
fi
timeout -k 10 600 python3 bench.py --steps 20 --warmup 2 --pmc-dump $out/pmc_rows.json > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -20 $out/bench.err; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 2 --cpu-baseline 0 --pmc off > $out/bench_under_rocprof.json 2> $out/stats.err || echo "stats run failed"
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
# (bench.py starts its own two ranks -- no launcher; the same command under torch.distributed.run is what the driver uses for N > 1)
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 4 --warmup 1 --envs-per-gpu 16384 > $out/bench_2rank_gloo.json 2> $out/bench_2rank.err || { echo "2-rank rehearsal failed"; tail -20 $out/bench_2rank.err; }
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 4 --warmup 1 --envs-per-gpu 16384 > $out/bench_2rank_gloo_torchrun.json 2> $out/bench_2rank_torchrun.err || { echo "2-rank rehearsal under torch.distributed.run failed"; tail -20 $out/bench_2rank_torchrun.err; }
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || echo "smoke failed"
tail -1 $out/smoke.txt
python3 - "$out" <<'PY'
import json, sys
o = sys.argv[1]
try:
    b = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
    r = b["roofline"]
    print("headline", b["value"], "env-steps/s", b["ms_per_step"], "ms; frac", r["frac"], "issue", r["valu_issue_frac"], "traffic", r["traffic"], "|", r["pmc_source"])
    for a in b.get("also", []):
        rr = a["roofline"]
        print(" ", a["workload"][:60], a["value"], a["unit"], "frac", rr["frac"], "issue", rr.get("valu_issue_frac"), "traffic", rr["traffic"])
    print("cpu", b.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
head -4 $out/kernel_stats.csv 2>/dev/null | cut -c1-200
tail -1 $out/bench_2rank_gloo.json 2>/dev/null | cut -c1-400
