"""Per-SIMD timeline of one step launch from the release library's placement table (stg_get_placement: SIMD, start and retire time of
every wavefront).  For each configuration: launch span, mean SIMD busy share, the tail, the spread of per-SIMD busy time and the
SIMDs that retire last with the wavefronts they held.   python3 tools/simd_timeline.py [row ...]   (rows: see ROWS)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench  # noqa: E402
import spin_torque_gym_amd as stg  # noqa: E402

bench.cap_host_threads()
ROWS = {
    "headline": dict(n=65536, solver="rk45", thermal=1),
    "rk4th": dict(n=65536, solver="rk4", thermal=1),
    "shard": dict(n=131072, solver="rk45", thermal=1),
    "cfg4": dict(n=262144, solver="rk4", thermal=0, mixed=True),
    "cfg4dev": dict(n=262144, solver="rk4", thermal=0, mixed=True, tm="device"),
    "rk4_262k": dict(n=262144, solver="rk4", thermal=0),
    "rk4th_262k": dict(n=262144, solver="rk4", thermal=1),
    "refill1m": dict(n=1048576, solver="rk45", thermal=1),
    "refill262k": dict(n=262144, solver="rk45", thermal=1),
    "cfg4perenv": dict(n=262144, solver="rk4", thermal=0, mixed=True, per_env=True),
    "rk4th_131k": dict(n=131072, solver="rk4", thermal=1),
    "rk4_131k": dict(n=131072, solver="rk4", thermal=0),
    "rk4_1m": dict(n=1048576, solver="rk4", thermal=0),
    "rk45_t0_131k": dict(n=131072, solver="rk45", thermal=0),
    "rk45_t0_65k": dict(n=65536, solver="rk45", thermal=0),
    "cfg2": dict(n=4096, solver="rk45", thermal=0),
    "hyb81920": dict(n=81920, solver="rk45", thermal=1),
    "hyb70000": dict(n=70000, solver="rk45", thermal=1),
}


def run(name, n, solver, thermal, mixed=False, tm="reference", steps=4, per_env=False):
    kw = dict(include_thermal_fluctuations=bool(thermal), temperature=300.0, solver=solver, seed=1234, autoreset=True, torque_model=tm)
    cls = None
    if mixed:
        mk, cls = bench.mixed_kwargs(solver, n, per_env)
        kw.update(mk)
    else:
        kw.update(device_params=bench.stt_params(bench.volume_for(solver)))
    env = stg.SpinTorqueVecEnv(n, class_index=cls, **kw)
    env.reset(seed=1234)
    b = env.backend
    acts = bench.make_actions(steps, n, b.device, 1234)
    for k in range(steps):
        b.step(acts[k], autoreset=True)
    torch.cuda.synchronize()
    p = b.placement(0, raw=True)
    where, prod, t0, t1, work, slot = p["where"], p["producer"], p["t0_us"], p["t1_us"], p["work"], p["slot"]
    sel = (~prod) & (t1 > 0)
    dur = t1 - t0
    print(f"== {name}: {n} envs {solver} thermal={thermal} {'mixed ' + tm if mixed else ''}: {p['workgroups']} workgroups x {p['waves_per_workgroup']} "
          f"wavefronts, {int(sel.sum())} integrating wavefronts recorded on {len(np.unique(where[sel]))} SIMDs")
    print(f"   span {p['span_us']:.1f} us, mean SIMD busy share {p['simd_busy_frac']:.3f}, tail after 90 % of the SIMDs retired {p['last_simd_alone_frac']:.3f}, "
          f"integrating wavefronts per SIMD {p['integrating_per_simd']}")
    start0 = t0[sel].min()
    keys = np.unique(where[sel])
    busy_sum = np.array([dur[sel & (where == k)].sum() for k in keys])
    last_end = np.array([t1[sel & (where == k)].max() - start0 for k in keys])
    print(f"   per-SIMD sum of wavefront durations: min {busy_sum.min():.1f} median {np.median(busy_sum):.1f} max {busy_sum.max():.1f} us; "
          f"per-SIMD last retire: min {last_end.min():.1f} median {np.median(last_end):.1f} max {last_end.max():.1f} us")
    print(f"   wavefront durations: min {dur[sel].min():.1f} median {np.median(dur[sel]):.1f} max {dur[sel].max():.1f} us; starts after launch: "
          f"median {np.median(t0[sel] - start0):.1f} max {(t0[sel] - start0).max():.1f} us")
    us_per_unit = dur[sel] / np.maximum(work[sel], 1)
    print(f"   us per work unit of a wavefront: min {us_per_unit.min():.3f} median {np.median(us_per_unit):.3f} max {us_per_unit.max():.3f}")
    order = np.argsort(-last_end)
    for k in list(keys[order[:3]]) + list(keys[order[-2:]]):
        m = sel & (where == k)
        iv = sorted(zip((t0[m] - start0).round(1).tolist(), (t1[m] - start0).round(1).tolist(), work[m].tolist(), slot[m].tolist()))
        co = prod & (where == k)
        print(f"   SIMD {int(k):#x} retires last at {float(max(b_ for _, b_, _, _ in iv)):.1f} us: integrating (start, retire, work units, wave slot) {iv}"
              + (f", producers {sorted(zip((t0[co] - start0).round(1), (t1[co] - start0).round(1)))}" if co.any() else ""))
    env.close()


for name in (sys.argv[1:] or list(ROWS)):
    run(name, **ROWS[name])
