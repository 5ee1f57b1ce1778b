"""Debug: per-step host wall time of the config-4 device-physics run as bench.py's `also` list executes it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
for tm in ("reference", "device", "device", "reference"):
    t_all = []
    import spin_torque_gym_amd as stg
    t0 = time.perf_counter()
    m = bench.run_config(262144, "rk4", 0, 8, 1, 0, 1, 0, mixed=True, torque_model=tm)
    print(tm, "wall per step %.3f ms, kernel %.3f ms, total call %.1f ms" % (m["wall_s"] / 8 * 1e3, m["kernel_ms_avg"], (time.perf_counter() - t0) * 1e3), flush=True)
