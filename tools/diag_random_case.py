"""Diagnosis helper: replays tests/test_gpu_parity.py::test_randomised_configurations_vs_oracle for one solver / sweep
offset and prints, per case, the deviation HIP vs oracle and (thermal) wave-specialised vs one-wavefront kernels.
python3 tools/diag_random_case.py <solver> <offset> <cases>"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "spin-torque-rl-gym_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import spin_torque_gym_amd as stg
from conftest import stt_default_params
from helpers import OracleBackend, unit_rows

solver, off, cases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng({"rk4": 101, "euler": 202, "rk45": 303}[solver] + 1000 * off)
n = 96
for case in range(cases):
    thermal = bool(case & 1)
    vol = float(10 ** rng.uniform(-11.5, -10) if solver != "rk45" else 10 ** rng.uniform(-5.7, -4.5))
    axis = np.array([0.0, 0.0, 1.0]) if case < 2 else np.array([rng.normal(0, 0.3), rng.normal(0, 0.3), 1.0])
    par = stt_default_params(volume=vol, damping=float(10 ** rng.uniform(-2.3, -0.5)),
                             saturation_magnetization=float(rng.uniform(4e5, 1.2e6)),
                             uniaxial_anisotropy=float(rng.uniform(3e5, 1.5e6)), easy_axis=axis,
                             polarization=float(rng.uniform(0.2, 0.9)))
    if case >= 4:
        par["demag_factors"] = np.array([0.1, 0.25, 0.65])
    tmax = 2e-10 if solver == "rk45" else 1.5e-9
    seed = 1000 + case
    r0 = np.random.default_rng(seed)
    m0 = unit_rows(r0, n)
    tgt = np.where(r0.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    res = {}
    for name, kw in (("hip", {}), ("hip_ws_off", dict(wave_spec=False)), ("oracle", dict(backend=OracleBackend))):
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, seed=seed, device_params=par, include_thermal_fluctuations=thermal, solver=solver,
                                   max_duration=5e-9, **kw)
        env.reset(options={"initial_state": m0, "target_state": tgt})
        arng = np.random.default_rng(seed + 1)
        ms = []
        for s in range(2):
            a = np.empty((n, 2), dtype=np.float32)
            a[:, 0] = arng.uniform(-2e6, 2e6, n) * (arng.uniform(0, 1, n) > 0.15)
            a[:, 1] = 10 ** arng.uniform(-12.2, np.log10(tmax), n)
            _, _, _, _, info = env.step(torch.from_numpy(a))
            ms.append((env.get_state()["m"].cpu().numpy().copy(), a.copy(), info["status"].cpu().numpy().copy()))
        res[name] = ms
        env.close()
    for s in range(2):
        d = np.abs(res["hip"][s][0] - res["oracle"][s][0]).max(axis=0) if res["hip"][s][0].shape[0] == 3 else np.abs(res["hip"][s][0] - res["oracle"][s][0]).max(axis=1)
        dw = np.abs(res["hip"][s][0] - res["hip_ws_off"][s][0]).max()
        j = int(np.argmax(d))
        print(f"case {case} thermal={thermal} step {s}: worst |dm| hip-oracle {d.max():.3e} at env {j} (J={res['hip'][s][1][j, 0]:.4g}, T={res['hip'][s][1][j, 1]:.4g}, "
              f"status {res['hip'][s][2][j]}); wave_spec on-off {dw:.1e}; alpha={par['damping']:.3g} vol={vol:.3g} Ms={par['saturation_magnetization']:.3g} Ku={par['uniaxial_anisotropy']:.3g}")
