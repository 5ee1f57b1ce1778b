"""One configuration of workload 2a (1 ps pulses) for profiling: python3 tools/run_2a_once.py <n> <K> [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import torch
import spin_torque_gym_amd as stg

n, K = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
env = stg.SpinTorqueVecEnv(n, solver="rk45", include_thermal_fluctuations=False, seed=1, autoreset=True)
env.reset(seed=0)
a = torch.zeros((K, n, 2), dtype=torch.float32, device="cuda")
a[..., 1] = 1e-12
at = a.transpose(1, 2).contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(reps + 2):
    if r == 2:
        e0.record()
    if K > 1:
        env.step_many(at, actions_soa=True, out_every=False)
    else:
        env.step(a[0])
e1.record()
torch.cuda.synchronize()
c = env.backend.counters()
print(f"n={n} K={K}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us/launch; work/env-step {c['work_units'] / max(c['env_steps'], 1):.2f}")
