"""A/B of the priority alternation between the two refill queues of a SIMD (STG_REFILL_FAIR=<bit>, 0 = off): kernel ms of the RK45 +
thermal refill launch at sizes with 2048 queues, plus the two queues' retire times on the SIMDs that finish last (placement table)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench  # noqa: E402

bench.cap_host_threads()
bench.DEFAULT_BLOCKS = 3
tag = "fair=" + os.environ.get("STG_REFILL_FAIR", "default")
for n, thermal in ((1048576, 1), (786432, 1), (1048576, 0)):
    m = bench.run_config(n, "rk45", thermal, 6, 2, 0, 1, 0)
    pl = m["placement"][-1]
    print(f"[{tag}] rk45 thermal={thermal} n={n}: kernel {m['kernel_ms_avg']:.3f} ms (min {m['kernel_ms_min']:.3f}); waves/SIMD {pl['integrating_per_simd']}, "
          f"busy {pl['simd_busy_frac']}, tail {pl['last_simd_alone_frac']}", flush=True)
import spin_torque_gym_amd as stg
env = stg.SpinTorqueVecEnv(1048576, device_params=bench.stt_params(9.7e-6), solver="rk45", include_thermal_fluctuations=True, seed=1, autoreset=True)
env.reset(seed=1)
acts = bench.make_actions(2, 1048576, env.backend.device, 3)
for k in range(2):
    env.backend.step(acts[k], autoreset=True)
p = env.backend.placement(0, raw=True)
where, t0, t1 = p["where"], p["t0_us"], p["t1_us"]
first = {}
for k in np.unique(where)[:6]:
    m_ = where == k
    print(f"[{tag}]   SIMD {int(k):#x}: queues retire at {sorted((t1[m_] - t0.min()).round(0).tolist())} us")
env.close()
