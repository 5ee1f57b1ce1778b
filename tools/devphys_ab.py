"""Device-physics torque model rows (mixed STT/SOT/VCMA envs, RK4) at several sizes -- kernel ms.  STG_NO_REGROUP=1 switches the
plan kernel's longest-first renumbering of a tile's four-block groups off (98 304 <= N <= 131 072).  python3 tools/devphys_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bench.cap_host_threads()
tag = "regroup off" if os.environ.get("STG_NO_REGROUP") else "regroup on "
for rep in range(reps):
    for n, per_env, thermal in ((65536, False, 0), (70000, False, 0), (98304, False, 0), (100000, False, 0), (110000, False, 0), (120000, False, 0), (131072, False, 0),
                                (131072, True, 0), (98304, False, 1), (110000, False, 1), (131072, False, 1), (262144, False, 0)):
        m = bench.run_config(n, "rk4", thermal, 8, 2, 0, 1, 0, mixed=True, torque_model="device", per_env=per_env)
        print(f"[{tag}] device physics n={n} per_env={int(per_env)} thermal={thermal}: kernel {m['kernel_ms_avg']:.4f} ms", flush=True)
