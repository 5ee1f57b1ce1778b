"""Experiment (library built with -DSTG_PROFILE_LOOP): where each integrating wavefront of one RK45 env-step ran, for how
long, and at which shader clock.  python3 tools/probe_wave_records.py <n> <thermal> [reps]"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import numpy as np
import torch
import spin_torque_gym_amd as stg
from spin_torque_gym_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
thermal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
par = stg.DeviceFactory().get_default_parameters("stt_mram")
par.update(volume=9.7e-6)
env = stg.SpinTorqueVecEnv(n, solver="rk45", include_thermal_fluctuations=thermal, device_params=par, seed=1234, autoreset=True)
env.reset(seed=0)
g = torch.Generator().manual_seed(3)
a = torch.zeros((n, 2), dtype=torch.float32)
a[:, 0] = (torch.rand(n, generator=g) * 2 - 1) * 2e6
a[:, 1] = torch.rand(n, generator=g) * 0.9e-9 + 0.1e-9
a = a.cuda()
lib = _lib.load()
lib.stg_debug_waves.restype = ctypes.c_int
wg_waves = 2 if (thermal and n <= 65536) else (4 if n >= 65536 else 1)
n_rec = min(8192, (n + 63) // 64 * (2 if wg_waves == 2 else 1))
quiet = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # steps before the reported ones (bench.py's steady state)
for rep in range(quiet + reps):
    a[:, 0] = (torch.rand(n, generator=g, device="cpu") * 2 - 1).cuda() * 2e6
    a[:, 1] = (torch.rand(n, generator=g, device="cpu") * 0.9e-9 + 0.1e-9).cuda()
    if rep < quiet:
        env.step(a)
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(a); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (ctypes.c_longlong * (n_rec * 6))()
    assert lib.stg_debug_waves(buf, n_rec) == 0
    r = np.frombuffer(buf, dtype=np.int64).reshape(n_rec, 6)
    r = r[r[:, 1] != 0]
    if wg_waves == 2:
        r = r[::1]
    ticks, real = (r[:, 1] - r[:, 0]).astype(float), (r[:, 3] - r[:, 2]).astype(float)
    t_first = r[:, 2].min()
    start_ms, end_ms = (r[:, 2] - t_first) * 1e-5, (r[:, 3] - t_first) * 1e-5
    clk = ticks / np.maximum(real, 1) * 100.0
    att = r[:, 5].astype(float)
    hw, xcc = r[:, 4] & 0xFFFFFFFF, (r[:, 4] >> 32) & 0xF
    simd_key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 3)
    per_simd = collections.Counter(simd_key.tolist())
    share = np.array([per_simd[k] for k in simd_key.tolist()])
    print(f"rep {rep}: step {ms:.3f} ms; {len(r)} integrating wavefronts on {len(per_simd)} SIMDs "
          f"(SIMDs holding 1/2/3+ of them: {sum(v == 1 for v in per_simd.values())}/{sum(v == 2 for v in per_simd.values())}/{sum(v >= 3 for v in per_simd.values())})")
    print(f"   start {start_ms.min():.3f}..{start_ms.max():.3f} ms, end {end_ms.min():.3f}..{end_ms.max():.3f} ms; clock MHz "
          f"min/median/max {clk.min():.0f}/{np.median(clk):.0f}/{clk.max():.0f}")
    order = np.argsort(-end_ms)[:6]
    for i in order:
        print(f"   late finisher: wave {i:5d} attempts {att[i]:6.0f}, {ticks[i] / max(att[i], 1):7.0f} ticks/attempt, "
              f"lifetime {end_ms[i] - start_ms[i]:.3f} ms, clock {clk[i]:.0f} MHz, integrating wavefronts on its SIMD: {share[i]}")
    big = att > 0.9 * att.max()
    print(f"   wavefronts with > 90 % of the max attempts: {big.sum()}, ticks/attempt median {np.median(ticks[big] / att[big]):.0f}, "
          f"by SIMD sharing: " + ", ".join(f"{k}: {np.median((ticks / np.maximum(att, 1))[big & (share == k)]):.0f}" for k in sorted(set(share[big].tolist()))))
