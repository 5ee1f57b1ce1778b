import os, sys
sys.path.insert(0, os.getcwd())
import bench
bench.cap_host_threads(); bench.DEFAULT_BLOCKS = 3
tag = "hyb_prio=" + os.environ.get("STG_HYB_PRIO", "default") + " min=" + os.environ.get("STG_HYBRID_MIN", "default")
for solver, sizes in (("rk45", (65600, 70000, 77777, 81920, 90112, 98304)), ("rk4", (65600, 70000, 81920, 90112, 98304))):
    for n in sizes:
        m = bench.run_config(n, solver, 1, 8, 2, 0, 1, 0)
        pl = m["placement"][-1]
        print(f"[{tag}] {solver} thermal n={n}: kernel {m['kernel_ms_avg']:.4f} ms (min {m['kernel_ms_min']:.4f}) wg {pl['workgroups']}x{pl['waves_per_workgroup']} busy {pl['simd_busy_frac']} tail {pl['last_simd_alone_frac']}", flush=True)
