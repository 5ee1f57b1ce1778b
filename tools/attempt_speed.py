#!/usr/bin/env python3
"""How long ONE RK45 attempt of an integrating wavefront takes, from the release build's placement table (per-wavefront start /
retire time and attempts of its slowest lane): lone wavefronts (64 ... 4096 envs: at most one per SIMD), with / without the thermal
field, inline normals against producer/consumer pairs.  All pulses 1 ns (every lane of a wavefront does about the same work).
usage (GPU box): python3 tools/attempt_speed.py [rk4|rk45] [n ...]   (rk4: per sub-step)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import spin_torque_gym_amd as stg  # noqa: E402


def run(n, thermal, wave_spec, steps=3, solver="rk45"):
    kw = dict(include_thermal_fluctuations=bool(thermal), temperature=300.0, solver=solver, seed=7, autoreset=True,
              wave_spec=wave_spec, device_params=bench.stt_params(bench.volume_for(solver)))
    env = stg.SpinTorqueVecEnv(n, device_index=0, **kw)
    be = env.backend
    env.reset(seed=3)
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.empty((steps, 2, n), dtype=torch.float32)
    a[:, 0] = (torch.rand((steps, n), generator=g) * 2 - 1) * 2e6
    a[:, 1] = 1e-9
    a = a.to(be.device)
    be.step(a[0], autoreset=True)
    torch.cuda.synchronize()
    rows = []
    for k in range(1, steps):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        be.step(a[k], autoreset=True)
        ev1.record()
        torch.cuda.synchronize()
        p = be.placement(0, raw=True)
        sel = ~p["producer"] & (p["work"] > 0)
        dur = (p["t1_us"] - p["t0_us"])[sel]
        work = p["work"][sel]
        per = dur / work
        rows.append((ev0.elapsed_time(ev1) * 1e3, dur.max(), work.max(), np.median(per), per.min(), per.max(), p["waves_per_workgroup"], int(sel.sum())))
    env.close()
    r = rows[-1]
    print("%s n=%6d thermal=%d wave_spec=%-5s: kernel %8.1f us, longest wavefront %8.1f us for %5d attempts; us per attempt: median %.3f min %.3f max %.3f "
          "(%d integrating wavefronts, %d per workgroup)" % (solver, n, thermal, wave_spec, r[0], r[1], r[2], r[3], r[4], r[5], r[7], r[6]), flush=True)


if __name__ == "__main__":
    solver = "rk45"
    if len(sys.argv) > 1 and sys.argv[1] in ("rk4", "rk45", "euler"):
        solver = sys.argv.pop(1)
    sizes = [int(x) for x in sys.argv[1:]] or [64, 4096, 65536]
    for n in sizes:
        for thermal, ws in ((0, False), (1, False), (1, True)):
            run(n, thermal, ws, solver=solver)
