#!/usr/bin/env python3
"""RK45 solver seam on pathological start rows (NaN / zero / inf / tiny / huge / non-unit), HIP against the oracle: success flags,
accepted points and the returned row.  usage (GPU box): python3 tools/rk45_pathological.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402,F401  (puts the package on the path)
import spin_torque_gym_amd as stg  # noqa: E402
from spin_torque_gym_amd.backend import EnvConfig, HipBackend  # noqa: E402
from helpers import OracleBackend  # noqa: E402
from conftest import stt_default_params  # noqa: E402

rows = np.array([[0.0, 0.0, 1.0], [0.6, 0.0, 0.8], [np.nan, 0.0, 1.0], [0.0, 0.0, 0.0], [np.inf, 0.0, 0.0], [1e-200, 0.0, 0.0],
                 [1e200, 0.0, 0.0], [3.0, 4.0, 0.0], [0.0, np.nan, np.nan], [-np.inf, np.inf, 1.0], [1e-13, 0.0, 0.0], [1e-11, 1e-11, 0.0]])
n = len(rows) * 2
m0 = np.concatenate([rows, rows]).T.copy()
J = np.concatenate([np.zeros(len(rows)), np.full(len(rows), 1.5e6)])
T = np.full(n, 2e-11)
table = [stg.flatten_params(stg.DeviceFactory().create_device("stt_mram", stt_default_params(volume=9.7e-6)))]
for thermal in (False, True):
    res = []
    for B in (HipBackend, OracleBackend):
        b = B(n, EnvConfig(diagnostics=True, solver="rk45", include_thermal_fluctuations=thermal, temperature=300.0, seed=5))
        b.set_params(table, None)
        out = b.solve(torch.tensor(m0), torch.tensor(J), torch.tensor(T))
        res.append({k: torch.as_tensor(v).cpu().numpy().copy() for k, v in out.items() if v is not None})
        b.close()
    h, o = res
    for i in range(n):
        same = (h["success"][i] == o["success"][i] and h["n_points"][i] == o["n_points"][i]
                and np.allclose(h["m_final"][:, i], o["m_final"][:, i], rtol=0, atol=1e-9, equal_nan=True))
        print("thermal=%d row %-28s J=%8.1e: hip ok=%d n=%4d m=%s | oracle ok=%d n=%4d m=%s %s" % (
            thermal, str(m0[:, i]), J[i], h["success"][i], h["n_points"][i], np.array2string(h["m_final"][:, i], precision=4),
            o["success"][i], o["n_points"][i], np.array2string(o["m_final"][:, i], precision=4), "" if same else "  <-- DIFFERENT"))
