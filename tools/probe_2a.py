"""Workload 2a: env-step whose pulse is one DP5 step of 1 ps -- the HBM-shaped end of the env kernel."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import torch
import spin_torque_gym_amd as stg

def run(n, solver, K, lane_sort, thermal=False):
    env = stg.SpinTorqueVecEnv(n, solver=solver, include_thermal_fluctuations=thermal, seed=1, autoreset=True, lane_sort=lane_sort)
    env.reset(seed=0)
    a = torch.zeros((K, n, 2), dtype=torch.float32, device="cuda"); a[..., 1] = 1e-12
    at = a.transpose(1, 2).contiguous()          # [K,2,N]
    for _ in range(2):
        env.step_many(at, actions_soa=True, out_every=False) if K > 1 else env.step(a[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        env.step_many(at, actions_soa=True, out_every=False) if K > 1 else env.step(a[0])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    c = env.backend.counters()
    bytes_alg = 160 * n * (1 if K == 1 else 1) + (8 * n * (K - 1) if K > 1 else 0)   # K fused: state read/written once, actions K times
    print(f"n={n} {solver} K={K} sort={lane_sort}: {ms*1e3:.1f} us/launch, {n*K/ms*1e3:.3e} env-steps/s, "
          f"alg HBM {bytes_alg/ms/1e6:.0f} GB/s ({bytes_alg/ms/1e6/8000:.3f} of 8 TB/s), work/env-step {c['work_units']/max(c['env_steps'],1):.2f}")
    env.close()

for n in (262144, 1048576, 4194304):
    for solver in ("rk45", "rk4"):
        for K, ls in ((1, None), (1, False), (8, False)):
            run(n, solver, K, ls)
