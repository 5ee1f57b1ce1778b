"""Does replaying one captured hipGraph per step (plan kernel + step kernel) beat eager launches for the short kernels?
python3 tools/graph_probe.py  -- wall ms per step over 200 steps, eager vs graph replay, RK4 T=0 at 4096 / 65536 envs."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spin-torque-rl-gym_amd")):
    sys.path.insert(0, p)
import spin_torque_gym_amd as stg  # noqa: E402

torch.set_num_threads(4)
par = stg.DeviceFactory().get_default_parameters("stt_mram")
par["volume"] = 8.75e-11
for n, solver in ((4096, "rk4"), (65536, "rk4"), (4096, "rk45")):
    env = stg.SpinTorqueVecEnv(n, device_params=par, include_thermal_fluctuations=False, solver=solver, seed=1, autoreset=True)
    env.reset(seed=0)
    b = env.backend
    g0 = torch.Generator().manual_seed(0)
    a = torch.empty((2, n))
    a[0] = (torch.rand(n, generator=g0) * 2 - 1) * 2e6
    a[1] = 1e-10 + torch.rand(n, generator=g0) * 9e-10
    a = a.to(b.device)
    steps = 200
    for _ in range(5):
        b.step(a, autoreset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.step(a, autoreset=True)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.step(a, autoreset=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            b.step(a, autoreset=True)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / steps * 1e3
    print(f"{solver} n={n}: eager {eager:.4f} ms/step, graph replay {graph:.4f} ms/step", flush=True)
    env.close()
