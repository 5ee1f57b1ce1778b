import sys, time
sys.path.insert(0, "spin-torque-rl-gym_amd"); sys.path.insert(0, "tests")
import numpy as np, torch
import spin_torque_gym_amd as stg
from conftest import stt_default_params
for solver in ("rk4", "rk45"):
    env = stg.SpinTorqueEnv(device_params=stt_default_params(volume=9.7e-6 if solver == "rk45" else 8.75e-11), solver=solver,
                            include_thermal_fluctuations=False, max_steps=100000)
    env.reset(seed=0)
    a = np.array([1e6, 1e-10], dtype=np.float32)
    for _ in range(20): env.step(a)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n): env.step(a)
    dt = (time.perf_counter() - t0) / n
    print(f"N=1 facade {solver}: {dt*1e6:.0f} us per env.step() (pulse 0.1 ns)")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): env.step(a)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    env.close()
