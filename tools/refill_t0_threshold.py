"""T = 0 K RK45: from which size is the global-queue refill launch faster than one env per lane?  (STG_REFILL_MIN=<first size of the
refill launch>)  usage: [STG_REFILL_MIN=65537] python3 tools/refill_t0_threshold.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
bench.cap_host_threads(); bench.DEFAULT_BLOCKS = 3
tag = "refill_min=" + os.environ.get("STG_REFILL_MIN", "default")
for n in (73728, 81920, 98304, 114688, 131072, 147456, 196608):
    m = bench.run_config(n, "rk45", 0, 6, 2, 0, 1, 0)
    pl = m["placement"][-1]
    print(f"[{tag}] rk45 T=0K n={n}: kernel {m['kernel_ms_avg']:.3f} ms (min {m['kernel_ms_min']:.3f}) wg {pl['workgroups']}x{pl['waves_per_workgroup']}", flush=True)
