"""How far up does the hybrid wave-specialised launch (pairs for the 2048 - blocks longest blocks) beat the global-queue refill launch?
RK45 / RK4 + thermal; STG_HYBRID_MIN = fewest pairs for which the hybrid is used, STG_REFILL_MIN = first size of the refill launch."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
bench.cap_host_threads(); bench.DEFAULT_BLOCKS = 3
tag = "hybrid_min=" + os.environ.get("STG_HYBRID_MIN", "768") + " refill_min=" + os.environ.get("STG_REFILL_MIN", "default")
solvers = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rk45", "rk4"]
for solver in solvers:
    for n in (81920, 86016, 90112, 98304, 106496, 114688, 122880, 131072):
        m = bench.run_config(n, solver, 1, 8, 2, 0, 1, 0)
        pl = m["placement"][-1]
        print(f"[{tag}] {solver} thermal n={n}: kernel {m['kernel_ms_avg']:.4f} ms (min {m['kernel_ms_min']:.4f}) wg {pl['workgroups']}x{pl['waves_per_workgroup']} busy {pl['simd_busy_frac']} tail {pl['last_simd_alone_frac']}", flush=True)
