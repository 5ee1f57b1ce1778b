"""cfg2a (1 048 576 envs, one 1 ps DP5 step per pulse) kernel ms, once per library build (STG_HIP_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import spin_torque_gym_amd as stg
bench.cap_host_threads()
n = 1048576
tag = os.path.basename(os.environ.get("STG_HIP_LIBRARY", "shipped")) + " refill=" + os.environ.get("STG_REFILL", "auto")
env = stg.SpinTorqueVecEnv(n, solver="rk45", include_thermal_fluctuations=False, seed=1, autoreset=True, device_index=0, lane_sort=False)
env.reset(seed=0)
b = env.backend
a = torch.zeros((2, n), dtype=torch.float32, device=b.device); a[1] = 1e-12
for _ in range(4):
    b.step(a, autoreset=True)
torch.cuda.synchronize()
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.step(a, autoreset=True)
    e1.record(); torch.cuda.synchronize()
    print(f"[{tag}] cfg2a K=1: {e0.elapsed_time(e1) / 20:.4f} ms per launch", flush=True)
