"""A/B of the output layouts (records vs soa) on the bench shapes: kernel time per step, alternating inside one process."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import numpy as np, torch
import spin_torque_gym_amd as stg
from bench import make_actions, stt_params, volume_for
for n, solver, thermal in ((65536, "rk4", 1), (65536, "rk45", 1), (262144, "rk4", 0), (4096, "rk45", 0)):
    res = {}
    for rep in range(3):
        for layout in ("soa", "records"):
            env = stg.SpinTorqueVecEnv(n, device_params=stt_params(volume_for(solver)), include_thermal_fluctuations=bool(thermal),
                                       solver=solver, seed=1234, autoreset=True, out_layout=layout)
            env.reset(seed=1234)
            b = env.backend
            acts = make_actions(10, n, b.device, 1234)
            for k in range(2):
                b.step(acts[k], autoreset=True)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            torch.cuda.synchronize()
            for k in range(8):
                ev[k][0].record(); b.step(acts[2 + k], autoreset=True); ev[k][1].record()
            torch.cuda.synchronize()
            res.setdefault(layout, []).append(float(np.mean([x.elapsed_time(y) for x, y in ev])))
            env.close()
    print(n, solver, "thermal", thermal, {k: [round(v, 4) for v in vs] for k, vs in res.items()}, flush=True)
