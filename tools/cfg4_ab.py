"""cfg4 rows (262144 mixed envs, RK4, T = 0 K) under an environment knob, e.g.
STG_SNAKE=1 python3 tools/cfg4_ab.py [reps] -- kernel ms of the class-table / device-physics / per-env rows, and RK4 at other sizes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bench.cap_host_threads()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("STG_") and k != "STG_HIP_LIBRARY")
for rep in range(reps):
    for name, n, mixed, tm, per_env, thermal in (("cfg4 class table", 262144, True, "reference", False, 0), ("cfg4 device physics", 262144, True, "device", False, 0),
                                                 ("cfg4 per-env", 262144, True, "reference", True, 0), ("rk4 T=0 homogeneous 262144", 262144, False, "reference", False, 0),
                                                 ("rk4 T=0 homogeneous 1048576", 1048576, False, "reference", False, 0), ("rk4 thermal 262144", 262144, False, "reference", False, 1),
                                                 ("rk4 T=0 131072", 131072, False, "reference", False, 0)):
        m = bench.run_config(n, "rk4", thermal, 8, 2, 0, 1, 0, mixed=mixed, torque_model=tm, per_env=per_env)
        print(f"[{tag}] {name}: kernel {m['kernel_ms_avg']:.4f} ms", flush=True)
