#!/usr/bin/env python3
"""Soak of the lane-refill kernel: random batch sizes, envs per lane, refill windows, class tables, thermal on/off, action
dtypes, bad actions and exhausted attempt budgets -- every output bit, the state and the on-device counters against the
one-env-per-lane launch.  usage: python3 tools/soak_refill.py [seed] [cases] [max_envs]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spin-torque-rl-gym_amd")):
    sys.path.insert(0, p)
import spin_torque_gym_amd as stg  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
max_envs = int(sys.argv[3]) if len(sys.argv) > 3 else 150000
rng = np.random.default_rng(seed)
p0 = stg.DeviceFactory().get_default_parameters("stt_mram")
p0["volume"] = 9.7e-6
bad = 0
t_start = time.time()
for case in range(cases):
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 9000), rng.integers(9000, max_envs)]))
    r = int(rng.integers(2, 10))
    chk = int(rng.choice([1, 2, 7, 16, 64, 200]))
    thermal = bool(rng.integers(0, 2))
    mixed = bool(rng.integers(0, 2))
    f64 = bool(rng.integers(0, 2))
    budget = int(rng.choice([200000, 200000, 60]))           # 60: most solves run out of attempts -> STATUS_NOOP
    sort = [None, False][int(rng.integers(0, 2))]
    layout = ["records", "soa"][int(rng.integers(0, 2))]
    skip = bool(rng.integers(0, 3) == 0)                     # skip_done (no auto-reset): finished envs are not stepped
    kw = dict(device_params=p0)
    cls = None
    if mixed:
        q = dict(p0, damping=0.03, polarization=0.4, easy_axis=np.array([0.1, 0.0, 1.0]))
        kw = dict(device_type=["stt_mram", "stt_mram", "stt_mram"], device_params=[p0, q, dict(p0, volume=2e-6)])
        cls = rng.integers(0, 3, n).astype(np.uint8)
    g = torch.Generator().manual_seed(1000 * seed + case)
    acts = []
    for k in range(2):
        a = torch.empty((n, 2), dtype=torch.float64 if f64 else torch.float32)
        a[:, 0] = (torch.rand(n, generator=g, dtype=torch.float64) * 2 - 1) * 2e6
        a[:, 1] = 1e-12 + torch.rand(n, generator=g, dtype=torch.float64) ** 3 * 3e-10
        ib = torch.rand(n, generator=g) < 0.02
        a[ib, 0] = float("nan")
        a[torch.rand(n, generator=g) < 0.01, 1] = float("inf")
        acts.append(a)
    outs = []
    for refill in (False, f"{r},{chk}"):
        if refill:
            os.environ["STG_REFILL"] = refill
        else:
            os.environ["STG_REFILL"] = "0"
        env = stg.SpinTorqueVecEnv(n, include_thermal_fluctuations=thermal, solver="rk45", seed=11 + case, autoreset=not skip, skip_done=skip,
                                   max_steps=1 if skip else 2,
                                   diagnostics=True, class_index=cls, wave_spec=False, lane_sort=sort, max_attempts=budget,
                                   out_layout=layout, **kw)
        env.reset(seed=case)
        rec = []
        for a in acts:
            o, rw, te, tr, info = env.step(a)
            rec.append([t.clone() for t in (o, rw, te, tr, info["status"], info["reward_f64"], info["energy"]) + ((info["final_obs"],) if not skip else ())])
        st = env.get_state()
        rec.append([st[k].clone() for k in ("m", "target", "total_energy", "step_count", "rng_step")])
        c = env.backend.counters()
        env.close()
        outs.append((rec, c))
    same = all(torch.equal(x, y) for s1, s2 in zip(outs[0][0], outs[1][0]) for x, y in zip(s1, s2)) and outs[0][1] == outs[1][1]
    noop = outs[0][1]["noop_steps"]
    print(f"case {case}: n={n} R={r} check={chk} thermal={thermal} mixed={mixed} f64={f64} budget={budget} sort={sort} {layout} skip_done={skip}: "
          f"{'identical' if same else 'DIFFERENT'} (noop steps {noop}) [{time.time() - t_start:.0f} s]", flush=True)
    bad += not same
print("soak:", "ok" if not bad else f"{bad} cases differ")
sys.exit(1 if bad else 0)
