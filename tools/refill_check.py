#!/usr/bin/env python3
"""Lane refill (stg_step_refill_kernel) against the one-env-per-lane kernel: every output bit and the state must agree.
usage: python3 tools/refill_check.py [R[,check]] [sizes...]   (STG_REFILL is read when a context is created)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spin-torque-rl-gym_amd")):
    sys.path.insert(0, p)
import spin_torque_gym_amd as stg  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "2"
sizes = [int(x) for x in sys.argv[2:]] or [131072, 262144, 70001, 4096 * 9 + 17]
p = stg.DeviceFactory().get_default_parameters("stt_mram")
p["volume"] = 9.7e-6


def run(n, refill, thermal, mixed, steps=2):
    if refill:
        os.environ["STG_REFILL"] = refill
    else:
        os.environ.pop("STG_REFILL", None)
    kw = dict(device_params=p)
    cls = None
    if mixed:
        q = dict(p, damping=0.02, polarization=0.5)
        kw = dict(device_type=["stt_mram", "stt_mram"], device_params=[p, q])
        cls = (np.arange(n) % 2).astype(np.uint8)
    env = stg.SpinTorqueVecEnv(n, include_thermal_fluctuations=thermal, solver="rk45", seed=7, autoreset=True, max_steps=2,
                               diagnostics=True, class_index=cls, wave_spec=False, **kw)
    env.reset(seed=3)
    g = torch.Generator().manual_seed(n)
    out = []
    for k in range(steps):
        a = torch.empty((n, 2))
        a[:, 0] = (torch.rand(n, generator=g) * 2 - 1) * 2e6
        a[:, 1] = 1e-10 + torch.rand(n, generator=g) * 4e-10
        o, r, te, tr, info = env.step(a)
        out.append([t.clone() for t in (o, r, te, tr, info["status"], info["reward_f64"], info["energy"], info["final_obs"])])
    st = env.get_state()
    out.append([st[k].clone() for k in ("m", "target", "total_energy", "step_count", "rng_step")])
    c = env.backend.counters()
    env.close()
    return out, c


bad = 0
for n in sizes:
    for thermal in (False, True):
        for mixed in (False, True):
            a, ca = run(n, None, thermal, mixed)
            b, cb = run(n, spec, thermal, mixed)
            same = all(torch.equal(x, y) for s1, s2 in zip(a, b) for x, y in zip(s1, s2)) and ca == cb
            print(f"n={n} thermal={thermal} mixed={mixed}: refill {spec} vs off -> {'identical' if same else 'DIFFERENT'}  counters {ca}", flush=True)
            bad += not same
sys.exit(1 if bad else 0)
