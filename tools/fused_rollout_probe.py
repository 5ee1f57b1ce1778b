"""Open-loop rollouts: K env-steps per launch (stg_step_many) on the headline configuration -- per env-step time against K = 1.
(Every lane keeps its env for the K steps: the launch ends with max over lanes of the SUM of K steps' work, not K times the max.)
usage (GPU box): python3 tools/fused_rollout_probe.py [n]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench  # noqa: E402
import spin_torque_gym_amd as stg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for solver, thermal in (("rk45", 1), ("rk4", 1)):
    for K in (1, 2, 4, 8, 16):
        env = stg.SpinTorqueVecEnv(n, include_thermal_fluctuations=bool(thermal), temperature=300.0, solver=solver, seed=1234, autoreset=True,
                                   device_params=bench.stt_params(bench.volume_for(solver)))
        env.reset(seed=1234)
        b = env.backend
        reps = max(2, 16 // K)
        acts = bench.make_actions(K * (reps + 1), n, b.device, 1234).reshape(reps + 1, K, 2, n)
        call = (lambda a: b.step(a[0], autoreset=True)) if K == 1 else (lambda a: b.step_many(a, out_every=False, autoreset=True))
        call(acts[0]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            call(acts[1 + r])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * K)
        print(f"{solver} thermal={thermal} n={n} K={K:2d}: {ms:.4f} ms per env-step ({n / ms * 1e3:.3e} env-steps/s)", flush=True)
        env.close()
