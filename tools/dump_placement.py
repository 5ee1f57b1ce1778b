"""Dump the raw placement table (SIMD, start, retire, work per wavefront, in workgroup order) of one step launch of a bench row to an
.npz for offline analysis.  python3 tools/dump_placement.py <row of tools/simd_timeline.py> <out.npz> [env knobs are inherited]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench  # noqa: E402
import spin_torque_gym_amd as stg  # noqa: E402

sys.argv = sys.argv[:1] + sys.argv[1:]
name, out = sys.argv[1], sys.argv[2]
ROWS = {
    "headline": dict(n=65536, solver="rk45", thermal=1), "rk4th": dict(n=65536, solver="rk4", thermal=1),
    "shard": dict(n=131072, solver="rk45", thermal=1), "cfg4": dict(n=262144, solver="rk4", thermal=0, mixed=True),
    "cfg4dev": dict(n=262144, solver="rk4", thermal=0, mixed=True, tm="device"), "rk4_262k": dict(n=262144, solver="rk4", thermal=0),
    "hyb81920": dict(n=81920, solver="rk45", thermal=1), "hyb70000": dict(n=70000, solver="rk45", thermal=1),
}
r = ROWS[name]
bench.cap_host_threads()
n, solver = r["n"], r["solver"]
kw = dict(include_thermal_fluctuations=bool(r["thermal"]), temperature=300.0, solver=solver, seed=1234, autoreset=True, torque_model=r.get("tm", "reference"))
cls = None
if r.get("mixed"):
    mk, cls = bench.mixed_kwargs(solver, n)
    kw.update(mk)
else:
    kw.update(device_params=bench.stt_params(bench.volume_for(solver)))
env = stg.SpinTorqueVecEnv(n, class_index=cls, **kw)
env.reset(seed=1234)
b = env.backend
acts = bench.make_actions(4, n, b.device, 1234)
ms = []
for k in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.step(acts[k], autoreset=True); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
p = b.placement(0, raw=True)
np.savez(out, where=p["where"], producer=p["producer"], t0=p["t0_us"], t1=p["t1_us"], work=p["work"], slot=p["slot"],
         workgroups=p["workgroups"], wpw=p["waves_per_workgroup"], step_ms=np.array(ms), dur_last_action=acts[3][1].cpu().numpy(),
         cls=(cls.numpy() if cls is not None else np.zeros(1)))
print(name, "step ms", [round(x, 3) for x in ms], "span", p["span_us"], "busy", p["simd_busy_frac"], "tail", p["last_simd_alone_frac"])
