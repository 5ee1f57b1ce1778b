"""The drop-in boundary exercised from plain C: examples/c_client.c (C99, gcc; no Python, no PyTorch, no HIP headers) drives
`stg_create -> stg_set_params -> stg_reset -> K x stg_step -> stg_get_state` through include/spintorque_hip.h alone -- the calls
that stand in for SpinTorqueEnv.reset / step (/root/reference/spin_torque_gym/envs/spin_torque_env.py:250-407) -- and writes its
inputs and outputs to a file.  The GPU test replays the same inputs through the CPU oracle and compares; the CPU test checks that
the client is built by `__graft_entry__.build()` and that everything it imports from the library is declared in the header."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import stt_default_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "examples", "_build", "c_client")


def _client():
    if not os.path.exists(CLIENT):
        import __graft_entry__
        __graft_entry__.build_c_client()
    return CLIENT


def test_c_client_builds_and_imports_only_declared_symbols():
    exe = _client()
    assert os.access(exe, os.X_OK)
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    used = sorted(set(re.findall(r"\b(stg_\w+)", und)))
    header = open(os.path.join(ROOT, "include", "spintorque_hip.h")).read()
    declared = set(re.findall(r"\b(stg_\w+)\s*\(", header))
    assert used and set(used) <= declared, (used, declared)
    for name in ("stg_create", "stg_set_params", "stg_reset", "stg_step", "stg_get_state", "stg_get_counters", "stg_destroy"):
        assert name in used
    assert "Py" not in und and "torch" not in und and "c10" not in und        # no Python, no PyTorch behind it


def _read(path):
    with open(path, "rb") as f:
        n, K, rk45, abi = np.fromfile(f, dtype=np.int64, count=4)
        out = {"n": int(n), "K": int(K), "rk45": int(rk45), "abi": int(abi)}
        out["m0"] = np.fromfile(f, dtype=np.float64, count=3 * n).reshape(3, n)
        out["target"] = np.fromfile(f, dtype=np.float64, count=3 * n).reshape(3, n)
        out["actions"] = np.fromfile(f, dtype=np.float32, count=K * 2 * n).reshape(K, 2, n)
        out["obs0"] = np.fromfile(f, dtype=np.float32, count=12 * n).reshape(12, n)
        steps = []
        for _ in range(K):
            s = {"obs": np.fromfile(f, dtype=np.float32, count=12 * n).reshape(12, n),
                 "reward": np.fromfile(f, dtype=np.float32, count=n),
                 "reward64": np.fromfile(f, dtype=np.float64, count=n),
                 "energy": np.fromfile(f, dtype=np.float64, count=n),
                 "term": np.fromfile(f, dtype=np.uint8, count=n), "trunc": np.fromfile(f, dtype=np.uint8, count=n),
                 "status": np.fromfile(f, dtype=np.uint8, count=n),
                 "m": np.fromfile(f, dtype=np.float64, count=3 * n).reshape(3, n),
                 "etot": np.fromfile(f, dtype=np.float64, count=n)}
            steps.append(s)
        assert f.read() == b""
    out["steps"] = steps
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("solver,n,K,tol", [("rk4", 1000, 3, 1e-10), ("rk45", 300, 2, 1e-8)])
def test_c_client_on_gpu_vs_oracle(tmp_path, oracle_mod, solver, n, K, tol):
    import torch
    import spin_torque_gym_amd as stg
    from helpers import OracleBackend
    path = str(tmp_path / "c_client.bin")
    r = subprocess.run([_client(), solver, str(n), str(K), path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = _read(path)
    assert (d["n"], d["K"], d["rk45"]) == (n, K, int(solver == "rk45"))
    m = re.search(r"(\d+) env-steps, (\d+) integrator work units, (\d+) no-op steps", r.stdout)
    assert m and int(m.group(1)) == n * K and int(m.group(2)) >= 10 * n * K
    # the same episode through the reference-shaped host code on the CPU oracle
    env = stg.SpinTorqueVecEnv(n, backend=OracleBackend, diagnostics=True, solver=solver, include_thermal_fluctuations=False,
                               device_params=stt_default_params(volume=9.7e-6 if solver == "rk45" else 8.75e-11))
    obs, _ = env.reset(options={"initial_state": d["m0"].T.copy(), "target_state": d["target"].T.copy()})
    assert np.allclose(d["obs0"].T, obs.cpu().numpy(), rtol=3e-7, atol=1e-12)
    switched = 0
    for k in range(K):
        o, rew, te, tr, info = env.step(torch.from_numpy(d["actions"][k].T.copy()))
        s = d["steps"][k]
        mo = env.get_state()["m"].cpu().numpy()
        mo = mo if mo.shape[0] == 3 else mo.T
        assert np.array_equal(s["status"], info["status"].cpu().numpy())
        assert np.array_equal(s["term"].astype(bool), te.cpu().numpy().astype(bool))
        assert np.array_equal(s["trunc"].astype(bool), tr.cpu().numpy().astype(bool))
        assert np.abs(s["m"] - mo).max() <= tol
        assert np.allclose(s["obs"].T, o.cpu().numpy(), rtol=3e-7, atol=10 * tol)
        assert np.allclose(s["reward64"], info["reward_f64"].cpu().numpy(), rtol=1e-10, atol=10 * tol)
        assert np.allclose(s["reward"], rew.cpu().numpy(), rtol=3e-7, atol=10 * tol)
        assert np.allclose(s["energy"], info["energy"].cpu().numpy(), rtol=max(1e-12, 10 * tol), atol=0)
        assert np.abs(np.linalg.norm(s["m"], axis=0) - 1.0).max() < 1e-12
        switched += int(s["term"].sum())
    assert switched > 0            # the pulses do switch some envs
    env.close()
