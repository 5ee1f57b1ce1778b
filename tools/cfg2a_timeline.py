import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
import spin_torque_gym_amd as stg
bench.cap_host_threads()
n = 1048576
env = stg.SpinTorqueVecEnv(n, solver="rk45", include_thermal_fluctuations=False, seed=1, autoreset=True, device_index=0, lane_sort=False)
env.reset(seed=0)
b = env.backend
a = torch.zeros((2, n), dtype=torch.float32, device=b.device); a[1] = 1e-12
for _ in range(4):
    b.step(a, autoreset=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.step(a, autoreset=True)
e1.record(); torch.cuda.synchronize()
print("ms per launch", e0.elapsed_time(e1) / 10, "knobs", os.environ.get("STG_REFILL"))
p = b.placement(0, raw=True)
t0, t1, work = p["t0_us"], p["t1_us"], p["work"]
print("waves", len(t0), "span", p["span_us"], "busy", p["simd_busy_frac"], "durations min/med/max", np.min(t1 - t0), np.median(t1 - t0), np.max(t1 - t0), "start max", (t0 - t0.min()).max(), "work min/med/max", work.min(), np.median(work), work.max())
c = b.counters(); print(c)
