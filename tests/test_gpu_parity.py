"""Parity tests proper: the HIP path, called through the C-ABI, against (a) the golden vectors recorded from the
Python reference and (b) the CPU oracle on seeded random batches.  Need a real MI355X: `pytest -m gpu`.

Tolerances.  north_star: m(t) within 1e-5 relative of the CPU reference on identical inputs (thermal off).  The
assertions below are tighter, at the level actually achieved: the kernels use FMA contraction and the device's
division/sqrt/pow, so they differ from NumPy by rounding only.
  RK4 final m ............ <= 1e-10 absolute (unit vectors, up to 5000 sub-steps)
  RK45 trajectories ...... <= 1e-8 absolute, identical accepted-point counts on the golden cases
  obs (fp32) ............. <= 2 ulp of fp32 relative + 1e-12
  reward (fp64) .......... <= 1e-10 relative
  integer/boolean outputs  exact (sub-step counts, success flags, terminated/truncated, status)
"""
import os

import numpy as np
import pytest
import torch

from conftest import sot_default_params, stt_default_params, vcma_default_params

pytestmark = pytest.mark.gpu

TOL_RK4 = 1e-10
TOL_RK45 = 1e-8


@pytest.fixture(scope="module")
def stg():
    import spin_torque_gym_amd as s
    assert torch.cuda.is_available(), "these tests need the GPU"
    return s


def _backend(stg, n, table, cls=None, **cfg):
    from spin_torque_gym_amd.backend import EnvConfig, HipBackend
    b = HipBackend(n, EnvConfig(**{"diagnostics": True, **cfg}))
    b.set_params(table, cls)
    return b


def _flat(stg, d, dev="stt_mram"):
    return stg.flatten_params(stg.DeviceFactory().create_device(dev, d))


# ------------------------------------------------------------------------------------------------
# solver level vs golden vectors
# ------------------------------------------------------------------------------------------------
def test_rk4_solver_vs_golden_g1_g2_g3(stg, golden):
    g1, g2, g3 = golden("G1_simple_rk4_relax"), golden("G2_simple_rk4_stt"), golden("G3_simple_degenerate")
    table = [_flat(stg, stt_default_params()), _flat(stg, stt_default_params(volume=8.75e-11)),
             _flat(stg, stt_default_params(volume=2e-11))]
    m0, J, T, cls, ref_m, ref_ok, ref_n = [], [], [], [], [], [], []
    for k in range(len(g1["T"])):
        m0.append(g1["m0"][g1["m0_index"][k]]); J.append(0.0); T.append(g1["T"][k]); cls.append(0)
        ref_m.append(g1["m_final"][k]); ref_ok.append(g1["success"][k]); ref_n.append(g1["n_steps"][k])
    for k in range(len(g2["T"])):
        m0.append(g2["m0"][g2["m0_index"][k]]); J.append(g2["J"][k]); T.append(g2["T"][k])
        cls.append(1 if g2["volume"][k] > 5e-11 else 2)
        ref_m.append(g2["m_final"][k]); ref_ok.append(g2["success"][k]); ref_n.append(g2["n_steps"][k])
    n_g3 = len(g3["J"])
    for k in range(n_g3):
        m0.append(g3["m0"][g3["m0_index"][k]]); J.append(g3["J"][k]); T.append(g3["T"][k]); cls.append(0)
        ref_m.append(g3["robust_m_last"][k]); ref_ok.append(g3["success"][k]); ref_n.append(-1)
    n = len(T)
    b = _backend(stg, n, table, torch.tensor(cls, dtype=torch.uint8), solver="rk4", include_thermal_fluctuations=False)
    out = b.solve(torch.tensor(np.array(m0).T.copy()), torch.tensor(J), torch.tensor(T))
    mf = out["m_final"].cpu().numpy().T
    ok = out["success"].cpu().numpy().astype(bool)
    npts = out["n_points"].cpu().numpy()
    ref_ok = np.array(ref_ok, dtype=bool)
    assert np.array_equal(ok, ref_ok), np.nonzero(ok != ref_ok)
    ref_n = np.array(ref_n)
    sel = ref_n >= 0
    assert np.array_equal(npts[sel], ref_n[sel])                     # H5
    err = np.abs(mf - np.array(ref_m))[ok]
    assert err.max() <= TOL_RK4, err.max()
    assert (~ok).sum() > 50                                          # the degenerate fixture is exercised (H3)
    # failed solves hand back the initial state (fallback result)
    assert np.array_equal(mf[~ok], np.array(m0)[~ok])
    b.close()


def test_rk4_trajectory_vs_golden(stg, golden):
    g1, g2 = golden("G1_simple_rk4_relax"), golden("G2_simple_rk4_stt")
    table = [_flat(stg, stt_default_params()), _flat(stg, stt_default_params(volume=float(g2["traj_volume"])))]
    m0 = np.array([g1["m0"][0], g1["m0"][29], g2["m0"][int(g2["traj_m0_index"])]]).T.copy()
    J = [0.0, 0.0, float(g2["traj_J"])]
    T = [1e-9, float(np.float32(3.3e-10)), float(g2["traj_T"])]
    b = _backend(stg, 3, table, torch.tensor([0, 0, 1], dtype=torch.uint8), solver="rk4", include_thermal_fluctuations=False)
    out = b.solve(torch.tensor(m0), torch.tensor(J, dtype=torch.float64), torch.tensor(T, dtype=torch.float64), traj_cap=1002)
    m = out["m"].cpu().numpy()       # [cap,3,N]
    t = out["t"].cpu().numpy()
    for lane, (rm, rt) in enumerate(((g1["traj0_m"], g1["traj0_t"]), (g1["traj1_m"], g1["traj1_t"]), (g2["traj_m"], g2["traj_t"]))):
        k = len(rt)
        assert int(out["n_points"][lane]) == k - 1
        assert np.abs(m[:k, :, lane] - rm).max() <= TOL_RK4
        assert np.abs(t[:k, lane] - rt).max() <= 1e-15 * rt[-1] + 1e-30
    b.close()


@pytest.mark.parametrize("thermal", [False, True])
def test_rk45_pathological_start_rows_vs_oracle(stg, thermal):
    """LLGSSolver.solve on start rows that are not unit vectors: NaN / zero / infinite components, a norm below the reference's
    1e-12 threshold, a non-unit row (llgs_solver.py:76,97-101).  Success flag, accepted points and the returned row equal the
    oracle's.  (The attempt loop evaluates the `|y| > 1e-12 else +z` test in the prologue only -- stg_physics.hpp: llgs_rhs -- this
    is the test that nothing observable hangs on it.)  One documented deviation: a finite row whose SQUARED norm overflows
    (|m0| > 1.3e154) fails here and comes back unchanged, where the reference divides by inf, integrates the zero vector and
    returns NaN (T = 0 K) or a noise-driven row (thermal) with success = True."""
    from helpers import OracleBackend
    from spin_torque_gym_amd.backend import EnvConfig, HipBackend
    rows = np.array([[0.0, 0.0, 1.0], [0.6, 0.0, 0.8], [np.nan, 0.0, 1.0], [0.0, 0.0, 0.0], [np.inf, 0.0, 0.0], [1e-200, 0.0, 0.0],
                     [3.0, 4.0, 0.0], [0.0, np.nan, np.nan], [-np.inf, np.inf, 1.0], [1e-13, 0.0, 0.0], [1e-11, 1e-11, 0.0],
                     [1e200, 0.0, 0.0]])
    k = len(rows)
    n = 2 * k
    m0 = np.concatenate([rows, rows]).T.copy()
    J = np.concatenate([np.zeros(k), np.full(k, 1.5e6)])
    T = np.full(n, 2e-11)
    table = [_flat(stg, stt_default_params(volume=9.7e-6))]
    res = []
    for B in (HipBackend, OracleBackend):
        b = B(n, EnvConfig(diagnostics=True, solver="rk45", include_thermal_fluctuations=thermal, temperature=300.0, seed=5))
        b.set_params(table, None)
        out = b.solve(torch.tensor(m0), torch.tensor(J), torch.tensor(T))
        res.append({key: torch.as_tensor(out[key]).cpu().numpy().copy() for key in ("m_final", "n_points", "success")})
        b.close()
    h, o = res
    overflow = np.arange(n) % k == k - 1
    sel = ~overflow
    assert np.array_equal(h["success"][sel], o["success"][sel])
    assert np.array_equal(h["n_points"][sel], o["n_points"][sel])
    assert np.allclose(h["m_final"][:, sel], o["m_final"][:, sel], rtol=0, atol=TOL_RK45, equal_nan=True)
    assert h["success"][sel].sum() == 2 * 5                        # the five rows with a usable direction, at both currents
    assert not h["success"][overflow].any() and np.array_equal(h["m_final"][:, overflow], m0[:, overflow])


@pytest.mark.parametrize("name,vols", [("G4_llgs_rk45_relax", {0: None}), ("G5_llgs_rk45_stt", {0: 9.7e-6, 1: 2e-6})])
def test_rk45_solver_vs_golden(stg, golden, name, vols):
    g = golden(name)
    cases = g["cases"]
    tags = sorted(vols)
    table = [_flat(stg, stt_default_params() if vols[t] is None else stt_default_params(volume=vols[t])) for t in tags]
    n = len(cases)
    cls = torch.tensor([tags.index(int(c[5])) for c in cases], dtype=torch.uint8)
    if len(table) == 1:
        cls = None
    b = _backend(stg, n, table, cls, solver="rk45", include_thermal_fluctuations=False)
    cap = max(len(g[f"t_{k}"]) for k in range(n)) + 8
    out = b.solve(torch.tensor(cases[:, :3].T.copy()), torch.tensor(cases[:, 4].copy()), torch.tensor(cases[:, 3].copy()),
                  traj_cap=cap, want_energy=True)
    t, m, e = out["t"].cpu().numpy(), out["m"].cpu().numpy(), out["energy"].cpu().numpy()
    for k in range(n):
        rt, rm, re = g[f"t_{k}"], g[f"m_{k}"], g[f"energy_{k}"]
        assert bool(out["success"][k]) == bool(cases[k, 6])
        assert int(out["n_points"][k]) == len(rt) - 1, (k, int(out["n_points"][k]), len(rt) - 1)
        kk = len(rt)
        assert np.abs(t[:kk, k] - rt).max() <= 1e-9 * rt[-1]
        assert np.abs(m[:kk, :, k] - rm).max() <= TOL_RK45
        assert np.abs(e[:kk, k] - re).max() <= 1e-8 * np.abs(re).max()
        assert np.abs(out["m_final"][:, k].cpu().numpy() - rm[-1]).max() <= TOL_RK45
    b.close()


# ------------------------------------------------------------------------------------------------
# env level vs golden episodes (through the drop-in SpinTorqueEnv facade, N = 1)
# ------------------------------------------------------------------------------------------------
def _episode_setup(tag):
    from test_oracle_golden import EPISODE_CFG, _episode_params
    dev, d = _episode_params(tag)
    return dev, d, EPISODE_CFG.get(tag, {})


def test_env_episodes_vs_golden_g6(stg, golden):
    g = golden("G6_env_episode")
    for k, tag in enumerate(g["episode_tags"]):
        tag = str(tag)
        dev, d, kw = _episode_setup(tag)
        env = stg.SpinTorqueEnv(device_type=dev, device_params=d, include_thermal_fluctuations=False, **kw)
        obs0, info0 = env.reset(seed=0, options={"initial_state": g[f"ep{k}_m0"], "target_state": g[f"ep{k}_target"]})
        assert np.allclose(obs0, g[f"ep{k}_obs"][0], rtol=2e-7, atol=1e-12), tag
        for j, a in enumerate(g[f"ep{k}_actions"]):
            obs, r, te, tr, info = env.step(np.array(a, dtype=np.float32))
            assert np.allclose(obs, g[f"ep{k}_obs"][j + 1], rtol=2e-7, atol=1e-12), (tag, j, obs, g[f"ep{k}_obs"][j + 1])
            rr = g[f"ep{k}_reward"][j]
            assert abs(r - rr) <= 1e-10 * max(1.0, abs(rr)), (tag, j, r, rr)
            assert te == bool(g[f"ep{k}_terminated"][j]) and tr == bool(g[f"ep{k}_truncated"][j]), (tag, j)
            assert info["simulation_success"] == bool(g[f"ep{k}_success"][j]), (tag, j)
            re_ = g[f"ep{k}_energy"][j]
            assert abs(info["energy_consumed"] - re_) <= 1e-12 * max(abs(re_), 1e-300), (tag, j)
            assert np.abs(env.current_magnetization - g[f"ep{k}_m"][j + 1]).max() <= TOL_RK4, (tag, j)
            assert abs(env.total_energy - g[f"ep{k}_total_energy"][j]) <= 1e-12 * max(abs(g[f"ep{k}_total_energy"][j]), 1e-300)
        env.close()


def test_cfg1_single_env_rk45_step_through_the_facade(stg, golden):
    """BASELINE cfg1 as worded -- "single STT-MRAM macrospin, SpinTorque-v0 default params, RK45, thermal off" -- through
    `SpinTorqueEnv(solver='rk45').step()`: reset with the initial state of every G4 relaxation (the
    unmodified reference's LLGSSolver.solve: default STT parameters, J = 0, T = 0.1 / 1 ns), ONE step((0, T)).
    (i) float64 action (T is the golden's exact double): the env's magnetisation equals the reference trajectory's last row to
    1e-8; (ii) float32 action (what the reference env casts to; T is then 2.8e-17 s off for 1 ns): observation / reward / flags /
    info equal the same facade on the oracle backend, whose C entry takes float32 actions."""
    from helpers import OracleBackend
    g = golden("G4_llgs_rk45_relax")
    worst = 0.0
    z = np.array([0.0, 0.0, 1.0])

    def one_step(backend, m0, action):
        env = stg.SpinTorqueEnv(include_thermal_fluctuations=False, solver="rk45", backend=backend)
        assert env.get_solver_info()["method"] == "rk45"
        obs0, _ = env.reset(seed=0, options={"initial_state": m0, "target_state": z})
        obs, r, te, tr, info = env.step(action)
        assert "error" not in info, info
        out = (obs0, obs, r, te, tr, info["simulation_success"], info["energy_consumed"], info["pulse_duration"],
               env.current_magnetization.copy(), env.step_count)
        env.close()
        return out
    for k, case in enumerate(g["cases"]):
        m0, T, J = case[:3], float(case[3]), float(case[4])
        assert J == 0.0
        h = one_step(None, m0, np.array([J, T], dtype=np.float64))
        assert h[7] == T and h[5] is True and h[9] == 1 and h[6] == 0.0
        d = np.abs(h[8] - g[f"m_{k}"][-1]).max()
        worst = max(worst, d)
        assert d <= TOL_RK45, (k, d)
        assert np.allclose(h[1][:3], g[f"m_{k}"][-1].astype(np.float32), rtol=3e-7, atol=1e-9)      # obs[:3] = m as float32
        h32, o32 = (one_step(b, m0, np.array([J, T], dtype=np.float32)) for b in (None, OracleBackend))
        assert np.array_equal(h32[0], o32[0]) and np.allclose(h32[1], o32[1], rtol=3e-7, atol=1e-9), k
        assert abs(h32[2] - o32[2]) <= 1e-8 and h32[3:8] == o32[3:8] and h32[9] == o32[9] == 1, (k, h32[2:8], o32[2:8])
        assert np.abs(h32[8] - o32[8]).max() <= TOL_RK45
        # the float32 duration's own effect on the end point: |dm/dt| <= gamma H_k ~ 5.3e11 /s times the rounding of T
        assert np.abs(h32[8] - g[f"m_{k}"][-1]).max() <= 1e-8 + 1.5 * 5.3e11 * abs(float(np.float32(T)) - T)
    print("cfg1: SpinTorqueEnv(solver='rk45').step vs the reference's G4 final rows: worst |dm| =", worst)


def test_reset_seed_parity_g9(stg, golden):
    """reset(seed=s) reproduces the reference's PCG64 draws (tests/integration/test_environment.py:77-93)."""
    g = golden("G9_reset_seeds")
    env = stg.SpinTorqueEnv(include_thermal_fluctuations=False)
    for s in range(64):
        obs, _ = env.reset(seed=s)
        assert np.abs(env.current_magnetization - g["state"][s, :3]).max() <= 1e-15
        assert np.array_equal(env.target_magnetization, g["state"][s, 3:])
        assert np.allclose(obs, g["obs"][s], rtol=2e-7, atol=0)
    env.reset(seed=1234)
    for row in g["continued_from_1234"]:
        env.reset()
        assert np.abs(env.current_magnetization - row[:3]).max() <= 1e-15
        assert np.array_equal(env.target_magnetization, row[3:])
    env.close()


# ------------------------------------------------------------------------------------------------
# batched env vs the oracle on seeded random inputs (same host code, two backends)
# ------------------------------------------------------------------------------------------------
def _run_pair(stg, n, steps, actions_fn, seed=0, **kw):
    from helpers import OracleBackend, unit_rows
    kw.setdefault("seed", seed)          # both envs key their device generator with the same seed
    rng = np.random.default_rng(seed)
    m0 = unit_rows(rng, n)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    envs = [stg.SpinTorqueVecEnv(n, diagnostics=True, **kw), stg.SpinTorqueVecEnv(n, diagnostics=True, backend=OracleBackend, **kw)]
    outs = []
    for env in envs:
        o, _ = env.reset(options={"initial_state": m0, "target_state": tgt})
        rec = [dict(obs=o.cpu().numpy().copy())]
        arng = np.random.default_rng(seed + 1)
        for s in range(steps):
            a = actions_fn(arng, n, s)
            o, r, te, tr, info = env.step(torch.from_numpy(a))
            rec.append(dict(obs=o.cpu().numpy().copy(), reward=info["reward_f64"].cpu().numpy().copy(),
                            term=te.cpu().numpy().copy(), trunc=tr.cpu().numpy().copy(),
                            status=info["status"].cpu().numpy().copy(), energy=info["energy"].cpu().numpy().copy(),
                            m=env.get_state()["m"].cpu().numpy().copy()))
        outs.append(rec)
        env.close()
    return outs


def _compare(outs, tol_m, obs_rtol=3e-7):
    hip, ora = outs
    worst = 0.0
    assert np.allclose(hip[0]["obs"], ora[0]["obs"], rtol=obs_rtol, atol=1e-12)
    for s in range(1, len(hip)):
        h, o = hip[s], ora[s]
        assert np.array_equal(h["status"], o["status"]), (s, np.nonzero(h["status"] != o["status"]))
        assert np.array_equal(h["term"], o["term"]) and np.array_equal(h["trunc"], o["trunc"]), s
        worst = max(worst, np.abs(h["m"] - o["m"]).max())
        assert np.abs(h["m"] - o["m"]).max() <= tol_m, (s, np.abs(h["m"] - o["m"]).max())
        assert np.allclose(h["obs"], o["obs"], rtol=obs_rtol, atol=max(1e-12, 10 * tol_m)), s
        assert np.allclose(h["reward"], o["reward"], rtol=1e-10, atol=max(1e-12, 10 * tol_m)), s
        assert np.allclose(h["energy"], o["energy"], rtol=max(1e-12, 10 * tol_m), atol=0), s
    return worst


def _uniform_actions(jmax, tlo, thi):
    def fn(rng, n, s):
        a = np.empty((n, 2), dtype=np.float32)
        a[:, 0] = rng.uniform(-jmax, jmax, n)
        a[:, 1] = rng.uniform(tlo, thi, n)
        return a
    return fn


def test_cfg2_rk4_4096_envs_vs_oracle(stg):
    """BASELINE config 2 shape: 4096 STT-MRAM envs, T = 0 K (thermal off), well-conditioned volume, random pulses."""
    outs = _run_pair(stg, 4096, 3, _uniform_actions(2e6, 1e-10, 1e-9), device_params=stt_default_params(volume=8.75e-11),
                     include_thermal_fluctuations=False, solver="rk4", seed=3)
    worst = _compare(outs, TOL_RK4)
    print("cfg2 rk4 worst |dm| =", worst)
    # the batch does switch some envs and finishes some episodes
    assert outs[0][-1]["term"].sum() > 0


def test_cfg2_rk45_1024_envs_vs_oracle(stg):
    outs = _run_pair(stg, 1024, 2, _uniform_actions(2e6, 1e-10, 6e-10), device_params=stt_default_params(volume=9.7e-6),
                     include_thermal_fluctuations=False, solver="rk45", seed=4)
    worst = _compare(outs, TOL_RK45)
    print("cfg2 rk45 worst |dm| =", worst)


def test_default_params_noop_regime_vs_oracle(stg):
    """Default STT parameters with random non-zero currents: almost every step is the H3 no-op; exact status parity."""
    outs = _run_pair(stg, 512, 2, _uniform_actions(2e6, 1e-10, 5e-10), include_thermal_fluctuations=False, solver="rk4", seed=5)
    _compare(outs, TOL_RK4)
    st = outs[0][1]["status"]
    assert (st == 1).mean() > 0.9


def test_thermal_on_same_philox_stream_vs_oracle(stg):
    """Thermal on (config 3 shape, reduced N): kernel and oracle key the same Philox stream, so trajectories agree to
    rounding plus the fp32 transcendental differences in the normals (~1e-6 relative on a 1e-9-relative field)."""
    outs = _run_pair(stg, 2048, 2, _uniform_actions(2e6, 1e-10, 4e-10), device_params=stt_default_params(volume=8.75e-11),
                     include_thermal_fluctuations=True, temperature=300.0, solver="rk4", seed=1234)
    _compare(outs, 1e-9)
    # at V = 1e-30 m^3 the field moves m by ~1e-4 per step: agreement to 1e-8 means the SAME normals were drawn
    # (an independent stream would differ at the 1e-4 level), up to the fp32 transcendental rounding of the normals
    def relax(rng, n, s):
        a = np.zeros((n, 2), dtype=np.float32); a[:, 1] = rng.uniform(1e-10, 3e-10, n); return a
    # (RK4 only: with RK45 the noise drives accept/reject decisions, so a 1e-7 difference in one normal can fork the
    # step sequence; its statistics are checked against the reference in test_g10_thermal_on_vs_reference)
    outs = _run_pair(stg, 512, 2, relax, device_params=stt_default_params(volume=1e-30), include_thermal_fluctuations=True,
                     temperature=300.0, solver="rk4", seed=4321)
    worst = _compare(outs, 2e-8, obs_rtol=1e-6)
    print("rk4 strong-noise same-stream worst |dm| =", worst)


def test_mixed_device_classes_vs_oracle(stg):
    """Config 4 shape (reduced N): STT/SOT/VCMA classes in one batch, per-class constants from LDS."""
    n = 3072
    cls = (np.arange(n) % 3).astype(np.uint8)
    types = ["stt_mram", "sot_mram", "vcma_mram"]
    params = [stt_default_params(volume=8.75e-11), sot_default_params(polarization=0.7, volume=8.75e-11),
              vcma_default_params(polarization=0.6, volume=5e-11, easy_axis=np.array([0.1, 0.0, 1.0]),
                                  reference_magnetization=np.array([0.0, 0.2, 1.0]))]
    outs = _run_pair(stg, n, 2, _uniform_actions(2e6, 1e-10, 5e-10), device_type=types, device_params=params,
                     class_index=cls, include_thermal_fluctuations=False, solver="rk4", seed=6)
    _compare(outs, TOL_RK4)


def test_bad_actions_and_truncation_vs_oracle(stg):
    def fn(rng, n, s):
        a = np.empty((n, 2), dtype=np.float32)
        a[:, 0] = rng.uniform(-3e6, 3e6, n)
        a[:, 1] = rng.uniform(-1e-9, 8e-10, n)
        a[::7, 0] = np.nan
        a[3::11, 1] = np.inf
        a[5::13, 0] = -np.inf
        return a
    outs = _run_pair(stg, 640, 4, fn, device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=False,
                     solver="rk4", max_steps=3, seed=7)
    _compare(outs, TOL_RK4)
    assert outs[0][3]["trunc"].all()


# ------------------------------------------------------------------------------------------------
# thermal field distribution (the reference's own moment test) and size-independent properties at full size
# ------------------------------------------------------------------------------------------------
def test_thermal_normals_moments_and_oracle_stream(stg, oracle_mod):
    b = _backend(stg, 4096, [_flat(stg, stt_default_params())], solver="rk4", seed=1234)
    z = b.thermal_normals(env_step=5, call0=0, n_calls=16).cpu().numpy()       # [16,3,4096]
    flat = z.transpose(1, 0, 2).reshape(3, -1)
    n = flat.shape[1]
    assert np.all(np.abs(flat.mean(axis=1)) < 5 / np.sqrt(n))
    assert np.all(np.abs(flat.std(axis=1) - 1.0) < 5 / np.sqrt(2 * n))
    # reference's tolerance on 1000 draws (tests/test_comprehensive_suite.py:447-476)
    sub = flat[:, :1000]
    assert np.all(np.abs(sub.mean(axis=1)) < 0.1) and np.all(np.abs(sub.std(axis=1) - 1) < 0.2)
    # kurtosis and cross-correlation sanity
    assert np.all(np.abs((flat ** 4).mean(axis=1) - 3.0) < 0.15)
    assert abs(np.corrcoef(flat)[0, 1]) < 0.01 and abs(np.corrcoef(flat)[0, 2]) < 0.01
    # same stream as the oracle (Philox4x32-10 + Box-Muller), up to fp32 transcendental rounding
    for env in (0, 1, 77, 4095):
        for call in (0, 3, 15):
            ref = oracle_mod.thermal_normals(1234, env, 5, call)
            assert np.abs(z[call, :, env] - ref).max() < 2e-5
    # strengths equal the reference's formulas (G8) through the library's host code
    assert np.isclose(b.thermal_strength(0), oracle_mod.thermal_strength(oracle_mod.make_params(stt_default_params()), 2.21e5, 300.0, 0), rtol=1e-15)
    b.close()


def test_full_size_properties_65536(stg):
    """Config 3 size: 65 536 envs, thermal on.  Size-independent properties: |m| = 1, finite outputs, determinism
    (same seed -> identical bits), partition invariance (two half-size contexts with env_id0 offsets = one full one)."""
    from spin_torque_gym_amd.backend import EnvConfig, HipBackend
    n = 65536
    rng = np.random.default_rng(11)
    v = rng.normal(0, 1, (3, n)); m0 = torch.tensor(v / np.linalg.norm(v, axis=0))
    tgt = torch.zeros((3, n), dtype=torch.float64); tgt[2] = torch.tensor(np.where(rng.integers(0, 2, n) == 0, 1.0, -1.0))
    act = torch.empty((2, n), dtype=torch.float32)
    act[0] = torch.tensor(rng.uniform(-2e6, 2e6, n), dtype=torch.float32)
    act[1] = torch.tensor(rng.uniform(1e-10, 3e-10, n), dtype=torch.float32)
    table = [_flat(stg, stt_default_params(volume=8.75e-11))]
    cfg = EnvConfig(solver="rk4", include_thermal_fluctuations=True, seed=99, diagnostics=True)

    def run(n_sub, off):
        b = HipBackend(n_sub, cfg, env_id0=off)
        b.set_params(table)
        b.reset(None, m0[:, off:off + n_sub], tgt[:, off:off + n_sub], 1)
        res = [t.clone() for t in b.step(act[:, off:off + n_sub])]
        m = b.get_state()["m"].clone()
        b.close()
        return res, m
    full, m_full = run(n, 0)
    again, m_again = run(n, 0)
    lo, m_lo = run(n // 2, 0)
    hi, m_hi = run(n // 2, n // 2)
    assert torch.equal(m_full, m_again) and all(torch.equal(a, b) for a, b in zip(full, again))
    assert torch.equal(m_full, torch.cat([m_lo, m_hi], dim=1))
    assert torch.equal(full[0], torch.cat([lo[0], hi[0]], dim=1))
    norm = torch.linalg.norm(m_full, dim=0)
    assert torch.all(torch.abs(norm - 1) < 1e-14)
    assert torch.isfinite(full[0]).all() and torch.isfinite(full[2]).all()


def test_skip_done_and_autoreset(stg):
    n = 256
    env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=False,
                               max_steps=2, skip_done=True, seed=5)
    env.reset(seed=1)
    a = torch.zeros((n, 2), dtype=torch.float32); a[:, 1] = 1e-10
    env.step(a); env.step(a)
    m_before = env.get_state()["m"].clone()
    _, _, te, tr, info = env.step(a)                      # every episode has ended by now: inactive, state frozen
    assert (te | tr).all() and (info["status"] == 3).all()
    assert torch.equal(env.get_state()["m"], m_before)
    env.close()
    env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=False,
                               max_steps=2, autoreset=True, seed=5)
    env.reset(seed=1)
    for _ in range(5):
        obs, r, te, tr, info = env.step(a)
    st = env.get_state()
    assert int(st["step_count"].max()) <= 2 and torch.all(torch.abs(torch.linalg.norm(st["m"], dim=0) - 1) < 1e-14)
    env.close()


def test_step_many_equals_repeated_step(stg):
    n, K = 512, 4
    rng = np.random.default_rng(2)
    acts = np.stack([_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, s) for s in range(K)])     # [K,N,2]
    kw = dict(device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=True, seed=8)
    e1 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw); e1.reset(seed=3)
    e2 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw); e2.reset(seed=3)
    obs_k = []
    for k in range(K):
        o, r, te, tr, info = e1.step(torch.from_numpy(acts[k]))
        obs_k.append((o.clone(), info["reward_f64"].clone(), te.clone(), tr.clone()))
    om, rm, tem, trm, infom = e2.step_many(torch.from_numpy(acts))
    for k in range(K):
        assert torch.equal(om[k], obs_k[k][0]) and torch.equal(infom["reward_f64"][k], obs_k[k][1])
        assert torch.equal(tem[k], obs_k[k][2]) and torch.equal(trm[k], obs_k[k][3])
    assert torch.equal(e1.get_state()["m"], e2.get_state()["m"])
    e1.close(); e2.close()


def test_state_dict_roundtrip(stg):
    n = 128
    kw = dict(device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=True, seed=8)
    a = torch.from_numpy(_uniform_actions(2e6, 1e-10, 3e-10)(np.random.default_rng(0), n, 0))
    e1 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw); e1.reset(seed=3); e1.step(a)
    sd = e1.state_dict()
    o1, r1, *_ = e1.step(a)
    e2 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw); e2.load_state_dict(sd)
    o2, r2, *_ = e2.step(a)
    assert torch.equal(o1, o2) and torch.equal(r1, r2)
    e1.close(); e2.close()


def test_solver_facades_on_gpu(stg, golden):
    """Reference-shaped solver classes (physics.py) on the HIP path: RobustLLGSSolver / LLGSSolver result dicts."""
    from spin_torque_gym_amd.physics import LLGSSolver, RobustLLGSSolver
    g1, g4 = golden("G1_simple_rk4_relax"), golden("G4_llgs_rk45_relax")
    params = stt_default_params()
    rs = RobustLLGSSolver(method="rk4", rtol=1e-3, atol=1e-6, timeout=2.0, max_retries=2, fallback_method="euler")
    r = rs.solve(g1["m0"][0], (0, 1e-9), params, lambda t: 0.0, lambda t: np.zeros(3), False, 300.0)
    assert r["success"] and np.abs(r["m"] - g1["traj0_m"]).max() <= TOL_RK4
    bad = rs.solve(g1["m0"][0], (0, 1e-9), params, lambda t: 1e6, None, False, 300.0)
    assert bad["success"] is False and bad.get("is_fallback")
    ls = LLGSSolver()
    c = g4["cases"][1]
    r = ls.solve(c[:3], (0, c[3]), params, lambda t: 0.0, None, thermal_noise=False)
    assert r["success"] and len(r["t"]) == len(g4["t_1"]) and np.abs(r["m"] - g4["m_1"]).max() <= TOL_RK45
    assert np.array_equal(r["torques"], g4["torques_1"])                             # J = 0: zeros
    # A8: |tau_stt| + |tau_fl| per accepted point, recorded on the device, against the reference's (G5, all cases)
    g5 = golden("G5_llgs_rk45_stt")
    for k, c5 in enumerate(g5["cases"]):
        r5 = ls.solve(c5[:3], (0, c5[3]), stt_default_params(volume={0: 9.7e-6, 1: 2e-6}[int(c5[5])]), lambda t: c5[4], None,
                      thermal_noise=False)
        tq = g5[f"torques_{k}"]
        assert len(r5["t"]) == len(tq) and np.abs(r5["torques"] - tq).max() <= 1e-8 * np.abs(tq).max(), k
        assert np.abs(r5["energy"] - g5[f"energy_{k}"]).max() <= 1e-8 * np.abs(g5[f"energy_{k}"]).max()
    st = ls.find_stable_states(params, n_trials=64, relax_time=2e-9, threshold=0.5, seed=0)
    assert 1 <= len(st) <= 2 and np.all(np.abs(np.abs(st[:, 2]) - 1.0) < 1e-2)      # relaxes to +-z
    # (f3) find_stable_states against the recorded reference runs (G18): the seeded global-np.random initial states, the
    # 10 ns relaxations (solve_batch on the recorded states) and the de-duplicated list
    from test_oracle_golden import G18_PARAMS
    g18 = golden("G18_stable_states")
    for tag in (str(t) for t in g18["tags"]):
        p18 = stt_default_params(**G18_PARAMS[tag])
        n18 = len(g18[f"{tag}_m_init"])
        rb = ls.solve_batch(g18[f"{tag}_m_init"], np.zeros(n18), np.full(n18, 10e-9), p18)
        assert rb["success"].all() and np.abs(rb["m_final"] - g18[f"{tag}_m_final"]).max() <= 1e-8
        np.random.seed(int(g18[f"{tag}_seed"]))
        st = ls.find_stable_states(p18, n_trials=n18)
        assert st.shape == g18[f"{tag}_stable_states"].shape and np.abs(st - g18[f"{tag}_stable_states"]).max() <= 1e-8


def test_float64_actions_match_reference_semantics(stg):
    """A float64 action array is clamped in float64 (monitoring.py:304-313 works in the array's dtype)."""
    n = 64
    kw = dict(device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=False, seed=2)
    e32 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw); e64 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw)
    e32.reset(seed=1); e64.reset(seed=1)
    a = torch.zeros((n, 2), dtype=torch.float64); a[:, 0] = 1.5e6; a[:, 1] = 1e-12      # 1e-12 is exact only in float64
    o64, *_ = e64.step(a)
    o32, *_ = e32.step(a.float())
    # float32(1e-12) < 1e-12 -> clipped up to exactly 1e-12 in both; identical physics, identical outputs
    assert torch.equal(o32, o64)
    a[:, 1] = 3.3e-10                                   # not representable in float32: the two runs see different T
    o64, *_ = e64.step(a); o32, *_ = e32.step(a.float())
    assert not torch.equal(o32[:, :3], o64[:, :3]) and torch.allclose(o32[:, :3], o64[:, :3], atol=1e-3)
    e32.close(); e64.close()


def test_g10_thermal_on_vs_reference(stg, golden):
    """Thermal field ON, HIP path against the reference's recorded samples (statistical; see test_oracle_golden)."""
    from test_oracle_golden import check_thermal_diffusion
    g = golden("G10_thermal_diffusion")
    m0 = g["rk4_m0"]
    table = [_flat(stg, stt_default_params(volume=float(v))) for v in sorted(set(g["wellcond"][:, 0]))]
    vols = sorted(set(g["wellcond"][:, 0]))
    rows = g["wellcond"]
    b = _backend(stg, len(rows), table, torch.tensor([vols.index(v) for v in rows[:, 0]], dtype=torch.uint8), solver="rk4",
                 include_thermal_fluctuations=True, seed=5)
    out = b.solve(torch.tensor(np.tile(m0, (len(rows), 1)).T.copy()), torch.tensor(rows[:, 1].copy()), torch.tensor(rows[:, 2].copy()))
    assert np.abs(out["m_final"].cpu().numpy().T - rows[:, 6:9]).max() < 1e-7
    assert np.array_equal(out["success"].cpu().numpy().astype(bool), rows[:, 9].astype(bool))
    b.close()

    def solve_many(solver, m0, T, vol, n):
        b = _backend(stg, n, [_flat(stg, stt_default_params(volume=vol))], solver=solver, include_thermal_fluctuations=True, seed=99)
        out = b.solve(torch.tensor(np.tile(m0, (n, 1)).T.copy()), torch.zeros(n, dtype=torch.float64),
                      torch.full((n,), T, dtype=torch.float64))
        res = out["m_final"].cpu().numpy().T.copy(), out["n_points"].cpu().numpy().copy()
        b.close()
        return res
    check_thermal_diffusion(g, solve_many, 65536, 16384, "hip")


def test_same_step_autoreset_vs_oracle(stg):
    """Auto-reset inside the step kernel (device-side draws from the env's Philox/xoshiro stream) vs its oracle
    restatement: obs of the new episodes, terminal observations, rewards and flags, over several episode boundaries."""
    outs = _run_pair(stg, 1024, 5, _uniform_actions(2e6, 1e-10, 3e-10), device_params=stt_default_params(volume=8.75e-11),
                     include_thermal_fluctuations=False, solver="rk4", max_steps=2, autoreset=True, seed=21)
    # the redrawn states come from fp32 Box-Muller normals: device transcendentals vs libm differ at the 1e-7 level,
    # and the switching dynamics amplifies that over the following steps (deterministic: fixed seeds, no flakiness)
    _compare(outs, 1e-3, obs_rtol=1e-3)
    hip, ora = outs
    fresh = hip[1]["term"].astype(bool)                       # episodes that ended (and were redrawn) at the first step
    assert fresh.any() and np.abs(hip[1]["m"][:, fresh] - ora[1]["m"][:, fresh]).max() < 2e-6
    assert outs[0][2]["trunc"].any() and outs[0][4]["trunc"].any()
    # after a reset the observation's step/energy/last-action fields are those of a fresh episode
    o = outs[0][2]["obs"][outs[0][2]["trunc"]]
    assert np.all(o[:, 8] == 1.0) and np.all(o[:, 9] == 0.0) and np.all(o[:, 10] == 0.0) and np.all(o[:, 11] == 0.0)


def test_harness_parity_single_env_rate_and_health(stg):
    """The reference's own gates on this path (tests/integration/test_environment.py:457-489: >= 10 steps/s;
    cli.py benchmark loop) and the health-report shape."""
    from spin_torque_gym_amd.harness import benchmark_physics_simulation, benchmark_vector_env
    r = benchmark_physics_simulation(steps=50, include_thermal_fluctuations=False)
    assert r["steps_per_second"] >= 10, r
    v = benchmark_vector_env(4096, steps=3, include_thermal_fluctuations=False)
    h = v["health"]
    assert set(h) >= {"timestamp", "health_status", "health_issues", "performance_metrics", "recent_performance"}
    assert h["performance_metrics"]["total_steps"] == 4096 * 5
    # default STT parameters + random currents: almost every solve fails and the report says so
    assert h["performance_metrics"]["solver_failure_rate"] > 0.9 and h["health_status"] == "WARNING"
    print("single-env steps/s", r["steps_per_second"], " vector env-steps/s", v["env_steps_per_second"])


@pytest.mark.parametrize("layout", ["soa", "records"])
def test_step_is_hip_graph_capturable(stg, layout):
    """stg_step enqueues only kernels on the caller's stream (no allocation, no synchronisation), so a caller can
    capture plan + step into a hipGraph and replay it; the replay gives the same bits as eager launches."""
    from spin_torque_gym_amd.backend import EnvConfig, HipBackend
    n = 8192
    table = [_flat(stg, stt_default_params(volume=8.75e-11))]
    cfg = EnvConfig(solver="rk4", include_thermal_fluctuations=True, seed=3, lane_sort=True, out_layout=layout, diagnostics=True)
    rng = np.random.default_rng(0)
    acts = torch.tensor(_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, 0).T.copy(), device="cuda")
    res = []
    for use_graph in (False, True):
        b = HipBackend(n, cfg); b.set_params(table); b.reset(None, None, None, 5)
        a_static = acts.clone()
        if use_graph:
            st0 = {k: v.clone() for k, v in b.get_state().items()}     # pristine state (counters at 0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                b.step(a_static)                                  # warm-up on the side stream (loads the code objects)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    b.step(a_static)
            b.set_state(st0); torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
        else:
            for _ in range(3):
                b.step(a_static)
        torch.cuda.synchronize()
        res.append((b.obs.clone(), b.get_state()["m"].clone(), b.counters()["env_steps"]))
        b.close()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_config5_shard_size_and_ragged_batches(stg):
    """Sizes: one config-5 shard (131 072 envs) and the full 1 048 576 on one GPU; ragged batches (N not a multiple of
    the wavefront, N = 1).  Size-independent properties: |m| = 1, counters add up, sub-batches equal the full batch."""
    from spin_torque_gym_amd.backend import EnvConfig, HipBackend
    table = [_flat(stg, stt_default_params(volume=8.75e-11))]
    cfg = EnvConfig(solver="rk4", include_thermal_fluctuations=True, seed=17, diagnostics=True)
    n = 1048576
    g = torch.Generator().manual_seed(1)
    v = torch.randn((3, n), generator=g, dtype=torch.float64)
    m0 = v / torch.linalg.norm(v, dim=0)
    tgt = torch.zeros((3, n), dtype=torch.float64); tgt[2] = 1.0
    act = torch.empty((2, n), dtype=torch.float32)
    act[0] = (torch.rand(n, generator=g) * 2 - 1) * 2e6
    act[1] = 1e-10 + torch.rand(n, generator=g) * 2e-10

    def run(lo, hi):
        b = HipBackend(hi - lo, cfg, env_id0=lo)
        b.set_params(table)
        b.reset(None, m0[:, lo:hi], tgt[:, lo:hi], 1)
        b.step(act[:, lo:hi])
        out = (b.obs.clone(), b.reward64.clone(), b.get_state()["m"].clone(), b.counters())
        b.close()
        return out
    obs, rew, m, c = run(0, n)
    assert c["env_steps"] == n and c["work_units"] >= 100 * n
    assert torch.all(torch.abs(torch.linalg.norm(m, dim=0) - 1) < 1e-14) and torch.isfinite(obs).all()
    for lo, hi in ((0, 1), (5, 68), (1000, 1000 + 131072), (n - 77, n)):       # N = 1, ragged, a config-5 shard, the tail
        o2, r2, m2, _ = run(lo, hi)
        assert torch.equal(o2, obs[:, lo:hi]) and torch.equal(r2, rew[lo:hi]) and torch.equal(m2, m[:, lo:hi])


def test_euler_solver_vs_golden_g11(stg, golden):
    g = golden("G11_simple_euler")
    vols = sorted(set(g["volume"]))
    table = [_flat(stg, stt_default_params(volume=float(v))) for v in vols]
    n = len(g["T"])
    b = _backend(stg, n, table, torch.tensor([vols.index(v) for v in g["volume"]], dtype=torch.uint8), solver="euler",
                 include_thermal_fluctuations=False)
    out = b.solve(torch.tensor(np.array([g["m0"][i] for i in g["m0_index"]]).T.copy()), torch.tensor(g["J"].copy()),
                  torch.tensor(g["T"].copy()))
    assert np.array_equal(out["success"].cpu().numpy().astype(bool), g["success"])
    assert np.array_equal(out["n_points"].cpu().numpy(), g["n_steps"])
    assert np.abs(out["m_final"].cpu().numpy().T - g["m_final"]).max() <= TOL_RK4
    b.close()


@pytest.mark.parametrize("solver", ["rk4", "euler", "rk45"])
def test_kernel_variant_matrix_vs_oracle(stg, solver):
    """Every template instantiation of the step kernel (solver x thermal x {one class, class table in LDS} x
    {easy axis = z specialisation, general axis} x {float32, float64 actions}) against the oracle, one step each."""
    from helpers import OracleBackend
    n = 192                                   # three wavefronts
    vol = 9.7e-6 if solver == "rk45" else 8.75e-11
    tilt = dict(easy_axis=np.array([0.15, -0.1, 1.0]), demag_factors=np.array([0.1, 0.2, 0.7]))
    rng = np.random.default_rng(77)
    from helpers import unit_rows
    m0 = unit_rows(rng, n)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    act = _uniform_actions(2e6, 1e-10, 3e-10)(rng, n, 0)
    worst = 0.0
    for thermal in (False, True):
        for multi in (False, True):
            for general_axis in (False, True):
                extra = tilt if general_axis else {}
                if multi:
                    kw = dict(device_type=["stt_mram", "vcma_mram"],
                              device_params=[stt_default_params(volume=vol, **extra),
                                             vcma_default_params(polarization=0.6, volume=vol * 0.8, **extra)],
                              class_index=(np.arange(n) % 2).astype(np.uint8))
                else:
                    kw = dict(device_params=stt_default_params(volume=vol, **extra))
                kw.update(include_thermal_fluctuations=thermal, solver=solver, seed=5)
                res = []
                for backend in (None, OracleBackend):
                    env = stg.SpinTorqueVecEnv(n, diagnostics=True, backend=backend, **kw)
                    env.reset(options={"initial_state": m0, "target_state": tgt})
                    o, r, te, tr, info = env.step(torch.from_numpy(act))
                    res.append((o.cpu().numpy().copy(), info["reward_f64"].cpu().numpy().copy(), te.cpu().numpy().copy(),
                                info["status"].cpu().numpy().copy(), env.get_state()["m"].cpu().numpy().copy()))
                    if backend is None:       # float64 actions carrying the same values take the other instantiation
                        env2 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw)
                        env2.reset(options={"initial_state": m0, "target_state": tgt})
                        o64, *_ = env2.step(torch.from_numpy(act.astype(np.float64)))
                        assert torch.equal(o64, o), (solver, thermal, multi, general_axis)
                        env2.close()
                    env.close()
                (o1, r1, t1, s1, m1), (o2, r2, t2, s2, m2) = res
                tag = (solver, thermal, multi, general_axis)
                tol = TOL_RK45 if solver == "rk45" else TOL_RK4
                assert np.array_equal(s1, s2) and np.array_equal(t1, t2), tag
                assert np.abs(m1 - m2).max() <= tol, (tag, np.abs(m1 - m2).max())
                assert np.allclose(o1, o2, rtol=3e-7, atol=1e-9) and np.allclose(r1, r2, rtol=1e-9, atol=1e-9), tag
                worst = max(worst, np.abs(m1 - m2).max())
    print(solver, "variant matrix worst |dm| =", worst)


@pytest.mark.parametrize("solver", ["rk4", "euler", "rk45"])
def test_wave_specialised_kernels_bit_identical(stg, solver):
    """The producer/consumer variant of the thermal kernels (a second wavefront runs the envs' normal streams ahead into
    LDS) draws the same values in the same order: every output must be bit-identical to the one-wavefront kernels, over
    ragged sizes, class tables, skip_done wavefronts that have nothing to integrate, fused steps and auto-reset."""
    n, K = 1000, 3                            # 15 full wavefronts + one ragged
    vol = 9.7e-6 if solver == "rk45" else 8.75e-11
    rng = np.random.default_rng(11)
    acts = np.stack([_uniform_actions(2e6, 1e-10, 4e-10)(rng, n, s) for s in range(K)])     # [K,N,2]
    acts[0, 5, 0] = np.nan                    # bad action in a wave of good ones
    acts[1, 64:128, 1] = 1e-12                # a wavefront of minimum-duration pulses
    for multi in (False, True):
        for mode in ("plain", "skip_done", "autoreset", "budget"):
            if mode == "budget" and solver != "rk45":
                continue                      # attempt budget: lanes give up mid-loop while their wavefront goes on
            if multi:
                kw = dict(device_type=["stt_mram", "vcma_mram"],
                          device_params=[stt_default_params(volume=vol),
                                         vcma_default_params(polarization=0.6, volume=vol * 0.8)],
                          class_index=(np.arange(n) % 2).astype(np.uint8))
            else:
                kw = dict(device_params=stt_default_params(volume=vol))
            kw.update(include_thermal_fluctuations=True, solver=solver, seed=21, max_steps=2 if mode != "plain" else 100,
                      skip_done=(mode == "skip_done"), autoreset=(mode == "autoreset"))
            if mode == "budget":
                kw.update(max_attempts=150)   # ~0.13 ns worth of attempts: most lanes end as STG_STATUS_NOOP
            outs = []
            for ws in (False, True):
                env = stg.SpinTorqueVecEnv(n, diagnostics=True, wave_spec=ws, **kw)
                env.reset(seed=4)
                o1, r1, te1, tr1, i1 = env.step(torch.from_numpy(acts[0]))
                om, rm, tem, trm, im = env.step_many(torch.from_numpy(acts))
                st = env.get_state()
                outs.append([o1.clone(), i1["reward_f64"].clone(), te1.clone(), i1["status"].clone(), om.clone(),
                             im["reward_f64"].clone(), tem.clone(), trm.clone(), st["m"].clone(), st["step_count"].clone()])
                env.close()
            for x, y in zip(*outs):
                assert torch.equal(x, y), (solver, multi, mode)
            if mode == "budget":
                noop = int((outs[0][3] == 1).sum())
                assert 0 < noop < n, noop


def test_g12_device_terms_kernel_vs_reference_formulas(stg, golden):
    from test_oracle_golden import G12_SOT, G12_VCMA
    g = golden("G12_device_terms")
    n = len(g["m"])
    for tag, over in G12_SOT.items():
        d = sot_default_params(**over)
        b = _backend(stg, n, [_flat(stg, d, "sot_mram")], solver="rk4")
        dl, fl, _ = b.device_terms(torch.tensor(g["m"].T.copy()), torch.tensor(g["J"].copy()), torch.zeros(n, dtype=torch.float64))
        # FMA contraction inside sigma x m: agreement to rounding of the vector's magnitude
        scale = np.abs(g[f"sot_{tag}_tau_dl"]).max()
        assert np.allclose(dl.cpu().numpy().T, g[f"sot_{tag}_tau_dl"], rtol=1e-14, atol=1e-15 * scale), tag
        assert np.allclose(fl.cpu().numpy().T, g[f"sot_{tag}_tau_fl"], rtol=1e-14, atol=0), tag
        b.close()
    nv = len(g["volts"])
    for tag, over in G12_VCMA.items():
        b = _backend(stg, nv, [_flat(stg, vcma_default_params(**over), "vcma_mram")], solver="rk4")
        _, _, ke = b.device_terms(torch.zeros((3, nv), dtype=torch.float64), torch.zeros(nv, dtype=torch.float64),
                                  torch.tensor(g["volts"].copy()))
        assert np.array_equal(ke.cpu().numpy(), g[f"vcma_{tag}_keff"]), tag
        b.close()


@pytest.mark.parametrize("thermal", [False, True])
def test_device_torque_model_mixed_batch_vs_oracle(stg, thermal):
    """BASELINE config 4 shape: mixed STT/SOT/VCMA batch with the opt-in device-physics torque terms (per-lane
    coefficients, type-grouped lane schedule) against the oracle's restatement; reference mode differs."""
    n = 3072
    cls = np.random.default_rng(3).integers(0, 3, n).astype(np.uint8)
    types = ["stt_mram", "sot_mram", "vcma_mram"]
    params = [stt_default_params(volume=8.75e-11),
              sot_default_params(polarization=0.7, volume=8.75e-11, current_direction=np.array([1.0, 0.3, 0.0])),
              vcma_default_params(polarization=0.6, volume=5e-11, vcma_coefficient=3e-13, easy_axis=np.array([0.1, 0.0, 1.0]))]
    common = dict(device_type=types, device_params=params, class_index=cls, include_thermal_fluctuations=thermal,
                  solver="rk4", seed=6)
    outs = _run_pair(stg, n, 2, _uniform_actions(2e6, 1e-10, 4e-10), torque_model="device", **common)
    _compare(outs, TOL_RK4 if not thermal else 1e-9)
    ref_mode = _run_pair(stg, n, 1, _uniform_actions(2e6, 1e-10, 4e-10), **common)
    dm = np.abs(outs[0][1]["m"] - ref_mode[0][1]["m"]).max(axis=0)
    assert np.all(dm[cls == 0] == 0.0)                       # STT lanes: identical bits in both models
    assert dm[cls == 1].max() > 1e-6 and dm[cls == 2].max() > 1e-6


# ------------------------------------------------------------------------------------------------
# SpinTorqueArray-v0
# ------------------------------------------------------------------------------------------------
def _array_kwargs(tag):
    from test_oracle_golden import G13_EPISODES, array_device_params
    ckw, coup, dev, over = G13_EPISODES[tag]
    kw = dict(array_size=(ckw["rows"], ckw["cols"]), action_mode=ckw["action_mode"], device_type=dev,
              device_params=array_device_params(dev, over) if (over or dev != "stt_mram") else None,
              observation_mode=ckw.get("obs_mode", "array"), include_coupling=ckw.get("include_coupling", True))
    for key in ("max_steps", "max_current", "max_duration", "success_threshold", "energy_penalty_weight", "temperature"):
        if key in ckw:
            kw[key] = ckw[key]
    if coup:
        kw.update(coupling_type=coup[0], coupling_strength=coup[1])
    return kw


def test_array_env_vs_golden_g13(stg, golden):
    """SpinTorqueArrayEnv on the HIP path against recorded reference episodes (all action modes, coupling types,
    observation modes, STT/SOT/VCMA cells)."""
    g = golden("G13_array_env")
    worst = 0.0
    for k, tag in enumerate(g["episode_tags"]):
        tag = str(tag)
        env = stg.SpinTorqueArrayEnv(**_array_kwargs(tag))
        obs, _ = env.reset(seed=k)
        assert np.abs(env.current_pattern - g[f"ep{k}_pattern"][0]).max() <= 1e-15, tag
        assert np.allclose(obs.reshape(-1), g[f"ep{k}_obs"][0], rtol=2e-7, atol=1e-12), tag
        for j, a in enumerate(g[f"ep{k}_actions"]):
            obs, r, te, tr, info = env.step(a)
            worst = max(worst, np.abs(env.current_pattern - g[f"ep{k}_pattern"][j + 1]).max())
            assert np.abs(env.current_pattern - g[f"ep{k}_pattern"][j + 1]).max() <= 1e-11, (tag, j)
            assert np.allclose(obs.reshape(-1), g[f"ep{k}_obs"][j + 1], rtol=3e-7, atol=1e-10), (tag, j)
            rr = g[f"ep{k}_reward"][j]
            assert abs(r - rr) <= 1e-9 * max(1.0, abs(rr)), (tag, j, r, rr)
            assert te == bool(g[f"ep{k}_terminated"][j]) and tr == bool(g[f"ep{k}_truncated"][j]), (tag, j)
            ee = g[f"ep{k}_energy"][j]
            assert abs(info["energy_consumed"] - ee) <= 1e-10 * max(abs(ee), 1e-300), (tag, j)
        env.close()
    print("array env worst |dm| vs reference =", worst)


def test_array_env_dict_observation(stg, golden):
    """observation_mode='dict': the facade against the recorded reference episode (G17), and the vector env's dict of
    tensors against its own 'vector' observation and device state."""
    from test_host_logic import _check_dict_episode_g17
    _check_dict_episode_g17(stg, golden, None, 1e-7)
    n = 256
    envs = {m: stg.SpinTorqueArrayVecEnv(n, (3, 4), observation_mode=m, max_steps=7, seed=5) for m in ("dict", "vector")}
    a = torch.zeros((n, 3), dtype=torch.float32)
    a[:, 0] = torch.arange(n) % 12; a[:, 1] = 1.5e6; a[:, 2] = 1e-9
    od, _ = envs["dict"].reset(seed=1)
    ov, _ = envs["vector"].reset(seed=1)
    for _ in range(2):
        od, rd, *_ = envs["dict"].step(a)
        ov, rv, *_ = envs["vector"].step(a)
    assert od["current_pattern"].shape == (n, 3, 4, 3) and od["steps_remaining"].dtype == torch.int64
    assert torch.equal(od["current_pattern"].reshape(n, -1), ov[:, :36]) and torch.equal(od["target_pattern"].reshape(n, -1), ov[:, 36:72])
    assert torch.equal(od["pattern_similarity"][:, 0], ov[:, 72]) and torch.equal(rd, rv)
    st = envs["dict"].get_state()
    assert torch.equal(od["steps_remaining"][:, 0], 7 - st["step_count"].to(torch.int64)) and int(od["steps_remaining"][0, 0]) == 5
    assert torch.equal(od["total_energy"][:, 0], st["total_energy"].to(torch.float32)) and float(od["total_energy"].min()) > 0
    for e in envs.values():
        e.close()


@pytest.mark.parametrize("mode", ["individual", "row", "column", "global"])
def test_array_vec_env_vs_oracle(stg, mode):
    from helpers import OracleArrayBackend
    n, size = 640, (4, 4)
    rng = np.random.default_rng(9)
    v = rng.normal(0, 1, (n, 4, 4, 3))
    init = v / np.linalg.norm(v, axis=-1, keepdims=True)
    hi = {"individual": 15, "row": 3, "column": 3}
    res = []
    for backend in (None, OracleArrayBackend):
        env = stg.SpinTorqueArrayVecEnv(n, size, action_mode=mode, coupling_type="dipolar", coupling_strength=0.2,
                                        success_threshold=0.05, max_steps=3, backend=backend)
        obs, _ = env.reset(options={"initial_pattern": init})
        rec = [obs.cpu().numpy().copy()]
        arng = np.random.default_rng(10)
        for s in range(4):
            if mode == "global":
                a = np.stack([arng.uniform(-2e6, 2e6, n), arng.uniform(1e-13, 1e-10, n)], axis=1).astype(np.float32)
            else:
                a = np.stack([arng.uniform(-1, hi[mode] + 1, n), arng.uniform(-3e6, 3e6, n), arng.uniform(-1e-10, 6e-9, n)],
                             axis=1).astype(np.float32)
                a[::9, 1] = 0.0
            obs, r, te, tr, info = env.step(torch.from_numpy(a))
            rec.append((obs.cpu().numpy().copy(), info["reward_f64"].cpu().numpy().copy(), te.cpu().numpy().copy(),
                        tr.cpu().numpy().copy(), info["energy"].cpu().numpy().copy(),
                        env.get_state()["pattern"].cpu().numpy().copy()))
        res.append(rec)
        env.close()
    hip, ora = res
    assert np.allclose(hip[0], ora[0], rtol=2e-7, atol=1e-12)
    for s in range(1, 5):
        assert np.abs(hip[s][5] - ora[s][5]).max() <= 1e-11, (mode, s)
        assert np.allclose(hip[s][0], ora[s][0], rtol=3e-7, atol=1e-10) and np.allclose(hip[s][1], ora[s][1], rtol=1e-9, atol=1e-9)
        assert np.array_equal(hip[s][2], ora[s][2]) and np.array_equal(hip[s][3], ora[s][3])
        assert np.allclose(hip[s][4], ora[s][4], rtol=1e-10, atol=0)
    assert hip[3][3].all() and hip[1][2].any()          # truncation at max_steps, some early successes


@pytest.mark.parametrize("device_type", ["stt_mram", "sot_mram"])
def test_array_global_mode_register_kernel_vs_lds_kernels_and_oracle(stg, device_type, monkeypatch):
    """4 x 4 arrays in 'global' mode run the kernel that keeps the pattern in (rotating) registers; STG_ARRAY_VARIANT=2 / 0 select the
    kernels that keep it in LDS (the 4 x 4 specialisation / the general one).  Same results to the rounding of the coupling sum's order
    (<= 1e-13 on the pattern after four steps), on a ragged batch, with and without coupling, in both observation modes, for a device
    type with shape demagnetisation too; and the oracle within the usual 1e-11."""
    from helpers import OracleArrayBackend
    n = 1000
    rng = np.random.default_rng(31)
    v = rng.normal(0, 1, (n, 4, 4, 3))
    init = v / np.linalg.norm(v, axis=-1, keepdims=True)
    acts = [np.stack([rng.uniform(-2e6, 2e6, n), rng.uniform(1e-13, 1e-10, n)], axis=1).astype(np.float32) for _ in range(4)]
    acts[1][::7, 1] = 0.0                                  # (action[1] is what 'global' mode reads as the current: some undriven arrays)
    for coupling in (True, False):
        for obs_mode in ("array", "vector"):
            runs = {}
            for name, variant, backend in (("registers", None, None), ("lds4x4", "2", None), ("general", "0", None), ("oracle", None, OracleArrayBackend)):
                if variant is None:
                    monkeypatch.delenv("STG_ARRAY_VARIANT", raising=False)
                else:
                    monkeypatch.setenv("STG_ARRAY_VARIANT", variant)
                env = stg.SpinTorqueArrayVecEnv(n, (4, 4), device_type=device_type, action_mode="global", include_coupling=coupling,
                                                coupling_type="dipolar", coupling_strength=0.2, observation_mode=obs_mode,
                                                success_threshold=0.05, max_steps=3, backend=backend)
                obs, _ = env.reset(options={"initial_pattern": init})
                rec = []
                for a in acts:
                    obs, r, te, tr, info = env.step(torch.from_numpy(a))
                    rec.append((obs.cpu().numpy().copy(), info["reward_f64"].cpu().numpy().copy(), te.cpu().numpy().copy(),
                                tr.cpu().numpy().copy(), info["energy"].cpu().numpy().copy(), env.get_state()["pattern"].cpu().numpy().copy()))
                runs[name] = rec
                env.close()
            monkeypatch.delenv("STG_ARRAY_VARIANT", raising=False)
            assert np.array_equal(runs["lds4x4"][-1][5], runs["general"][-1][5])            # (the two LDS kernels: same arithmetic)
            for s in range(4):
                reg = runs["registers"][s]
                for other, tol in (("lds4x4", 1e-13), ("oracle", 1e-11)):
                    o = runs[other][s]
                    assert np.abs(reg[5] - o[5]).max() <= tol, (coupling, obs_mode, other, s, np.abs(reg[5] - o[5]).max())
                    assert np.allclose(reg[0], o[0], rtol=3e-7, atol=1e-10) and np.allclose(reg[1], o[1], rtol=1e-9, atol=1e-9)
                    assert np.array_equal(reg[2], o[2]) and np.array_equal(reg[3], o[3])
                    assert np.allclose(reg[4], o[4], rtol=1e-10, atol=0)
            if not coupling:       # no coupling sum: nothing is reordered
                assert np.array_equal(runs["registers"][-1][5], runs["lds4x4"][-1][5])


def test_array_env_sizes_and_device_reset(stg):
    """8 x 8 cells (LDS: 98 KB pattern + 32 KB coupling per wavefront), device-side random reset, ragged batch."""
    n = 1000
    env = stg.SpinTorqueArrayVecEnv(n, (8, 8), action_mode="global", seed=4)
    obs, _ = env.reset(seed=2)
    st = env.get_state()
    pat = st["pattern"].reshape(64, 3, n)
    assert torch.all(torch.abs(torch.linalg.norm(pat, dim=1) - 1) < 1e-12)
    tgt = st["target"].reshape(64, 3, n)[:, 2, 0].reshape(8, 8).cpu().numpy()
    assert np.array_equal(tgt, np.where((np.add.outer(np.arange(8), np.arange(8)) % 2) == 0, 1.0, -1.0))
    a = torch.zeros((n, 2)); a[:, 1] = 1.0
    obs, r, te, tr, info = env.step(a)
    assert tuple(obs.shape) == (n, 64 * 6) and torch.isfinite(obs).all() and torch.isfinite(r).all()
    env.close()


def test_integration_md_ctypes_stub_runs(stg):
    """The ctypes stub printed in INTEGRATION.md section 3 runs as written (only the library path is substituted)."""
    import os
    import re
    from conftest import ROOT
    from spin_torque_gym_amd import _lib
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = [b for b in re.findall(r"```python\n(.*?)```", txt, flags=re.S) if "stg_create" in b][0]
    block = block.replace('C.CDLL("libspintorque_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(block, ns)
    torch.cuda.synchronize()
    obs = ns["obs"]
    assert torch.isfinite(obs).all() and torch.all(torch.abs(torch.linalg.norm(obs[:3].double(), dim=0) - 1) < 1e-6)


def test_pipelined_gather_overlaps_and_matches_sync(stg):
    """ShardedSpinTorqueVecEnv through RCCL with a world of one rank (the 8-GPU run is the driver's): the step kernel
    writes 56-byte records straight into the collective's send buffer (or, in-place / p2p, into this rank's slice of the
    global record array), the all-gather runs on its own stream under the next step's kernel (gather_begin/gather_end,
    two buffer pairs alternating), and the learner gets typed strided views -- no staging copy, no torch.cat.  Pipelined == synchronous == the plain single-GPU env, bit for
    bit, for the all-gather and for the point-to-point exchange."""
    import socket
    import torch.distributed as dist
    from spin_torque_gym_amd.distributed import ShardedSpinTorqueVecEnv
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, steps = 8192, 5
        rng = np.random.default_rng(3)
        acts = [torch.from_numpy(_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, k)) for k in range(steps)]
        kw = dict(device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=True, seed=9, autoreset=True)
        plain = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw); plain.reset(seed=2)
        want = []
        for a in acts:
            o, r, te, tr, _ = plain.step(a)
            want.append((o.clone(), r.clone(), te.clone(), tr.clone()))
        plain.close()
        for algo, inplace in (("all_gather", False), ("all_gather", True), ("p2p", False)):
            e1 = ShardedSpinTorqueVecEnv(n, gather_algo=algo, inplace=inplace, **kw); e1.reset(seed=2, gather=False)
            e2 = ShardedSpinTorqueVecEnv(n, gather_algo=algo, inplace=inplace, **kw); e2.reset(seed=2, gather=False)
            sync = []
            for a in acts:
                o, r, te, tr, _ = e1.step(a)
                assert tuple(o.shape) == (n, 12) and tuple(o.stride()) == (14, 1) and o.dtype == torch.float32
                assert te.dtype == torch.bool and r.dtype == torch.float32
                glob = [g.untyped_storage().data_ptr() for g in e1._glob]
                assert all(t.untyped_storage().data_ptr() in glob for t in (o, r, te, tr))     # views, not copies
                sync.append((o.clone(), r.clone(), te.clone(), tr.clone()))
            piped = []
            for k, a in enumerate(acts):
                e2.step(a, gather=False)
                if k:
                    piped.append(tuple(t.clone() for t in e2.gather_end()))     # step k-1, gathered under step k's kernel
                e2.gather_begin()
            piped.append(tuple(t.clone() for t in e2.gather_end()))
            for a, b, c in zip(sync, piped, want):
                assert all(torch.equal(x, y) for x, y in zip(a, b)), algo
                assert all(torch.equal(x, y) for x, y in zip(a, c)), algo
            e1.close(); e2.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("solver,thermal", [("rk4", True), ("rk45", False)])
def test_record_output_layout_equals_soa(stg, solver, thermal):
    """cfg.out_layout = STG_OUT_RECORDS (56-byte env-major records: obs[12] f32 | reward f32 | terminated | truncated |
    status) against the separate component-major arrays: same bits, for reset (incl. a mask), step with auto-reset,
    fused steps, a caller-provided record array, sorted and identity lane schedules, ragged sizes."""
    from spin_torque_gym_amd.backend import record_views
    vol = 9.7e-6 if solver == "rk45" else 8.75e-11
    for n in (1000, 70001):
        rng = np.random.default_rng(n)
        acts = np.stack([_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, k) for k in range(3)])
        kw = dict(device_params=stt_default_params(volume=vol), include_thermal_fluctuations=thermal, solver=solver, seed=4,
                  autoreset=True, max_steps=2)
        outs = []
        for layout in ("soa", "records"):
            env = stg.SpinTorqueVecEnv(n, diagnostics=True, out_layout=layout, **kw)
            o0, _ = env.reset(seed=1)
            rec = [o0.clone()]
            o, r, te, tr, info = env.step(torch.from_numpy(acts[0]))
            rec += [o.clone(), r.clone(), te.clone(), tr.clone(), info["status"].clone(), info["final_obs"].clone()]
            if layout == "records":
                assert tuple(o.stride()) == (14, 1)
                buf = torch.full((n, 56), 255, dtype=torch.uint8, device="cuda")
                o2, r2, te2, tr2, info2 = env.step(torch.from_numpy(acts[1]), out=buf)
                assert o2.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()
                assert torch.equal(record_views(buf)[4], info2["status"]) and bool((buf[:, 55] == 0).all())
            else:
                o2, r2, te2, tr2, info2 = env.step(torch.from_numpy(acts[1]))
            rec += [o2.clone(), r2.clone(), te2.clone(), tr2.clone()]
            om, rm, tem, trm, im = env.step_many(torch.from_numpy(acts))
            rec += [om.clone(), rm.clone(), tem.clone(), trm.clone(), im["reward_f64"].clone()]
            mask = torch.arange(n) % 3 == 0
            om2, _ = env.reset(seed=5, options={"mask": mask})
            rec += [om2.clone(), env.get_state()["m"].clone()]
            outs.append(rec)
            env.close()
        for j, (x, y) in enumerate(zip(*outs)):
            assert torch.equal(x, y), (solver, n, j)


def test_masked_reset_keeps_reward_and_flags_of_other_envs(stg):
    """with out_layout='records' the reward / terminated / truncated tensors step() returns are views of the
    record array a masked reset() writes the observations into.  The usual non-autoreset loop
    `o, r, te, tr, _ = env.step(a); env.reset(options={'mask': te | tr})` must not change r / te / tr of the envs it does not
    reset (the reset ones read 0, as the header says); in both layouts the observation rows of the others are their current
    observation with last_action = 0."""
    n = 4096
    rng = np.random.default_rng(8)
    a = _uniform_actions(2e6, 1e-10, 3e-10)(rng, n, 0)
    for layout in ("records", "soa"):
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=False,
                                   solver="rk4", seed=3, max_steps=2, out_layout=layout)
        env.reset(seed=1)
        env.step(torch.from_numpy(a))
        o, r, te, tr, info = env.step(torch.from_numpy(a))                   # max_steps = 2: every env is truncated now
        assert bool(tr.all()) and bool((r != 0).any())
        r0, te0, tr0, st0 = r.clone(), te.clone(), tr.clone(), info["status"].clone()
        o0 = o.clone()
        mask = torch.arange(n, device=r.device) % 3 == 0
        o1, _ = env.reset(seed=2, options={"mask": mask})
        keep = ~mask
        if layout == "records":
            from spin_torque_gym_amd.backend import record_views
            _, rr, tt, uu, ss = record_views(env.backend.packed)
            assert torch.equal(rr[keep], r0[keep]) and torch.equal(tt[keep].bool(), te0[keep]) and torch.equal(uu[keep].bool(), tr0[keep])
            assert torch.equal(ss[keep], st0[keep])
            assert bool((rr[mask] == 0).all()) and not bool(tt[mask].any()) and not bool(uu[mask].any())
            # the tensors handed out by step() are those views
            assert torch.equal(r[keep], r0[keep]) and torch.equal(tr[keep], tr0[keep])
        # observation of an env that was not reset: unchanged but for the last-action fields (reported as 0)
        assert torch.equal(o1[keep][:, :10], o0[keep][:, :10]) and bool((o1[keep][:, 10:] == 0).all())
        assert bool((o1[mask][:, 8] == 1.0).all())                           # fresh episodes: all steps remaining
        env.close()


@pytest.mark.parametrize("layout", ["records", "soa"])
def test_diagnostics_off_is_the_same_step_with_fewer_outputs(stg, layout):
    """VERDICT r2 item 4 / ADVICE r3: by default (diagnostics=False) a step passes NULL for the C-ABI's optional DIAGNOSTIC
    outputs (fp64 reward, energy, status array) and writes the RL-facing outputs only; obs / reward / flags / state are
    bit-identical to diagnostics=True, the status still rides in byte 54 of each record, and info carries no fp64 extras.
    The terminal observations of same-step auto-reset are RL-facing: info['final_obs'] is there with diagnostics off too and
    holds the same bits on every env whose episode ended (a learner bootstrapping at truncation reads it)."""
    n = 5000
    rng = np.random.default_rng(5)
    acts = [_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, k) for k in range(3)]
    kw = dict(device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=True, solver="rk4", seed=9,
              autoreset=True, max_steps=2, out_layout=layout)
    res = []
    for diag in (True, False):
        env = stg.SpinTorqueVecEnv(n, diagnostics=diag, **kw) if diag else stg.SpinTorqueVecEnv(n, **kw)
        assert env.backend.diagnostics is diag
        env.reset(seed=1)
        out = []
        for a in acts:
            o, r, te, tr, info = env.step(torch.from_numpy(a))
            if diag:
                assert {"status", "reward_f64", "energy", "final_obs"} <= set(info)
            else:
                assert set(info) == ({"status", "final_obs"} if layout == "records" else {"final_obs"})
                assert env.backend.reward64 is None and env.backend.energy is None
            ended = te | tr
            fo = info["final_obs"]
            assert tuple(fo.shape) == (n, 12)
            out.append((o.clone(), r.clone(), te.clone(), tr.clone(), info["status"].clone() if "status" in info else None,
                        fo[ended].clone(), ended.clone()))
            # an env that ended: its obs row is the NEW episode's first observation (all steps remaining, no last action), the
            # terminal one -- of a truncated env: no steps remaining -- is in final_obs
            assert bool((o[ended][:, 8] == 1.0).all()) and bool((o[ended][:, 10:] == 0.0).all())
            assert bool((fo[tr][:, 8] == 0.0).all()) and bool((fo[ended][:, 11] != 0.0).all())
        # (max_steps = 2: on the second step every env that did not reach its target on the first is truncated)
        assert float(out[1][6].float().mean()) > 0.5
        om, rm, tem, trm, im = env.step_many(torch.from_numpy(np.stack(acts)))
        assert ("reward_f64" in im) is diag and tuple(im["final_obs"].shape) == (3, n, 12)
        endm = tem | trm
        out.append((om.clone(), rm.clone(), tem.clone(), trm.clone(), im["status"].clone() if "status" in im else None,
                    im["final_obs"][endm].clone(), endm.clone()))
        out.append(env.get_state()["m"].clone())
        res.append(out)
        env.close()
    for k in range(len(acts) + 1):
        for x, y in zip(res[0][k][:4], res[1][k][:4]):
            assert torch.equal(x, y), (layout, k)
        if layout == "records":
            assert torch.equal(res[0][k][4], res[1][k][4])
        assert torch.equal(res[0][k][5], res[1][k][5]) and torch.equal(res[0][k][6], res[1][k][6]), (layout, k, "final_obs")
    assert torch.equal(res[0][-1], res[1][-1])


def test_misaligned_record_pointers_are_rejected(stg):
    """the records layout stores float pairs; a record array or final_obs that is not 8-byte aligned is an
    STG_E_INVALID from stg_reset / stg_step_many, not a GPU fault."""
    import ctypes as C
    from spin_torque_gym_amd import _lib
    from spin_torque_gym_amd.backend import EnvConfig
    lib = _lib.load()
    n = 64
    cfg = EnvConfig(solver="rk4", out_layout="records").to_abi()
    ctx = C.c_void_p()
    assert lib.stg_create(C.byref(ctx), 0, n, 0, C.byref(cfg)) == 0
    p = stg.devices.flatten_params(stg.DeviceFactory().create_device("stt_mram", stt_default_params()))
    assert lib.stg_set_params(ctx, C.byref(p), 1, None) == 0
    dev = torch.device("cuda", 0)
    raw = torch.zeros(n * 56 + 64, dtype=torch.uint8, device=dev)
    fin = torch.zeros(n * 48 + 64, dtype=torch.uint8, device=dev)
    act = torch.zeros((2, n), dtype=torch.float32, device=dev); act[1] = 1e-10
    assert raw.data_ptr() % 8 == 0
    good, bad = C.c_void_p(raw.data_ptr()), C.c_void_p(raw.data_ptr() + 4)
    assert lib.stg_reset(ctx, None, None, None, C.c_uint64(0), bad, None) == _lib.STG_E_INVALID and b"aligned" in lib.stg_last_error()
    assert lib.stg_reset(ctx, None, None, None, C.c_uint64(0), good, None) == 0
    a = C.c_void_p(act.data_ptr())
    assert lib.stg_step_many(ctx, 1, a, 0, 1, 1, bad, None, None, None, None, None, None, None, None) == _lib.STG_E_INVALID
    assert lib.stg_step_many(ctx, 1, a, 0, 1, 1, good, C.c_void_p(fin.data_ptr() + 4), None, None, None, None, None, None, None) == _lib.STG_E_INVALID
    assert b"final_obs" in lib.stg_last_error()
    assert lib.stg_step_many(ctx, 1, a, 0, 1, 1, good, C.c_void_p(fin.data_ptr()), None, None, None, None, None, None, None) == 0
    torch.cuda.synchronize()
    lib.stg_destroy(ctx)


@pytest.mark.parametrize("solver", ["rk4", "euler"])
def test_ornstein_uhlenbeck_noise_model_vs_oracle(stg, solver):
    """noise_model='ou' (ThermalFluctuations' correlated field) on the fixed-step kernels against the
    oracle, same stream: one- and two-wavefront kernels, one class and a class table, two fused steps; RK45 refuses it."""
    from helpers import OracleBackend, unit_rows
    n = 320
    rng = np.random.default_rng(31)
    m0 = unit_rows(rng, n)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    acts = [_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, k) for k in range(2)]
    vol = 1e-27                                  # strong field: the noise is most of the dynamics
    for multi in (False, True):
        if multi:
            kw = dict(device_type=["stt_mram", "vcma_mram"],
                      device_params=[stt_default_params(volume=vol), vcma_default_params(polarization=0.6, volume=vol * 0.8)],
                      class_index=(np.arange(n) % 2).astype(np.uint8))
        else:
            kw = dict(device_params=stt_default_params(volume=vol))
        kw.update(include_thermal_fluctuations=True, solver=solver, seed=13, noise_model="ou", correlation_time=3e-12,
                  max_current=0.0 + 2e6)
        res = []
        for backend, ws in ((None, False), (None, True), (OracleBackend, None)):
            env = stg.SpinTorqueVecEnv(n, diagnostics=True, backend=backend, wave_spec=ws, **kw)
            env.reset(options={"initial_state": m0, "target_state": tgt})
            out = []
            for a in acts:
                o, r, te, tr, info = env.step(torch.from_numpy(a * np.array([0.0, 1.0], dtype=np.float32)))   # J = 0: relaxation + noise
                out.append((env.get_state()["m"].cpu().numpy().copy(), info["status"].cpu().numpy().copy()))
            res.append(out)
            env.close()
        for k in range(2):
            if not np.array_equal(res[0][k][0], res[1][k][0]):
                dd = np.abs(res[0][k][0] - res[1][k][0]); bad = np.nonzero(dd.max(axis=0))[0]
                print("DIAG", solver, multi, k, "n_bad", len(bad), "lanes", bad[:16], "max", dd.max(), "oracle diff off/on",
                      np.abs(res[0][k][0] - res[2][k][0]).max(), np.abs(res[1][k][0] - res[2][k][0]).max())
            assert np.array_equal(res[0][k][0], res[1][k][0])                       # wave_spec on/off: bit-identical
            assert np.array_equal(res[0][k][1], res[2][k][1])
            d = np.abs(res[0][k][0] - res[2][k][0]).max()
            assert d <= 1e-8, (solver, multi, k, d)     # the normals carry fp32 device transcendentals (1e-7 relative)
        # the correlated field is not the white one
        envw = stg.SpinTorqueVecEnv(n, diagnostics=True, **{**kw, "noise_model": "white"})
        envw.reset(options={"initial_state": m0, "target_state": tgt})
        envw.step(torch.from_numpy(acts[0] * np.array([0.0, 1.0], dtype=np.float32)))
        assert np.abs(envw.get_state()["m"].cpu().numpy() - res[0][0][0]).max() > 1e-6
        envw.close()
    with pytest.raises(Exception):
        stg.SpinTorqueVecEnv(8, diagnostics=True, solver="rk45", noise_model="ou")


@pytest.mark.parametrize("solver", ["rk4", "euler", "rk45"])
def test_randomised_configurations_vs_oracle(stg, solver):
    """Randomised cross-check: device parameters drawn inside the validator's ranges (incl. tilted easy axes, demag
    factors, damping up to 0.3), log-uniform float32 pulse durations from the 1 ps minimum upwards (the roundings behind
    H4/H5 and the RK45 pulse gate), currents from 0 to the limit, thermal off and on, two steps each -- HIP vs oracle."""
    # STG_RANDOM_SWEEP=<offset>[,<cases>] widens the sweep by hand (other parameter draws, more cases); default: fixed.
    # (Thermal cases with a very small volume can exceed the tolerance there: the device's fp32 log/sin/cos differ from
    # libm's by an ulp, and a strong field over hundreds of Euler sub-steps amplifies that to ~1e-7 -- the
    # wave-specialised and one-wavefront kernels stay bit-identical; tools/diag_random_case.py shows both.)
    sweep = [int(x) for x in os.environ.get("STG_RANDOM_SWEEP", "0,6").split(",")]
    rng = np.random.default_rng({"rk4": 101, "euler": 202, "rk45": 303}[solver] + 1000 * sweep[0])
    n = 96
    worst = 0.0
    for case in range(sweep[1] if len(sweep) > 1 else 6):
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

        def actions(arng, k, s, tmax=tmax):
            a = np.empty((k, 2), dtype=np.float32)
            a[:, 0] = arng.uniform(-2e6, 2e6, k) * (arng.uniform(0, 1, k) > 0.15)          # some exact zeros
            a[:, 1] = 10 ** arng.uniform(-12.2, np.log10(tmax), k)                        # below the 1 ps clamp too
            return a
        outs = _run_pair(stg, n, 2, actions, seed=1000 + case, device_params=par, include_thermal_fluctuations=thermal,
                         solver=solver, max_duration=5e-9)
        tol = (TOL_RK45 if solver == "rk45" else TOL_RK4) * (50 if thermal else 1)       # normals carry fp32 device transcendentals
        worst = max(worst, _compare(outs, tol))
    print(solver, "randomised configurations: worst |dm| =", worst)


def test_c_abi_error_behaviour(stg):
    """Every entry point returns a negative STG_E_* code with a message instead of faulting: wrong call order, NULL
    pointers, invalid sizes and configurations (the reference's step() never raises for bad actions either, but its
    constructors do for bad configurations)."""
    import ctypes as C
    from spin_torque_gym_amd import _lib
    from spin_torque_gym_amd.backend import EnvConfig
    lib = _lib.load()
    cfg = EnvConfig(solver="rk4").to_abi()
    ctx = C.c_void_p()
    assert lib.stg_create(C.byref(ctx), 0, 0, 0, C.byref(cfg)) < 0 and b"n_envs" in lib.stg_last_error()
    assert lib.stg_create(C.byref(ctx), 99, 64, 0, C.byref(cfg)) < 0
    bad = EnvConfig(solver="rk4").to_abi(); bad.max_step = 0.0
    assert lib.stg_create(C.byref(ctx), 0, 64, 0, C.byref(bad)) < 0
    bad = EnvConfig(solver="rk45").to_abi(); bad.noise_model = 1
    assert lib.stg_create(C.byref(ctx), 0, 64, 0, C.byref(bad)) < 0 and b"fixed-step" in lib.stg_last_error()
    for field, val in (("lane_refill", 1), ("lane_refill", -2), ("lane_refill", 5000), ("reserved0", 7)):     # ABI v3
        bad = EnvConfig(solver="rk45").to_abi(); setattr(bad, field, val)
        assert lib.stg_create(C.byref(ctx), 0, 64, 0, C.byref(bad)) == _lib.STG_E_INVALID and field.encode() in lib.stg_last_error()
    assert lib.stg_create(C.byref(ctx), 0, 64, 0, C.byref(cfg)) == 0
    n = 64
    dev = torch.device("cuda", 0)
    obs = torch.empty((12, n), dtype=torch.float32, device=dev)
    rew = torch.empty(n, dtype=torch.float32, device=dev)
    te = torch.empty(n, dtype=torch.uint8, device=dev); tr = torch.empty_like(te)
    act = torch.zeros((2, n), dtype=torch.float32, device=dev); act[1] = 1e-10
    ptr = lambda t: C.c_void_p(t.data_ptr())
    # order: params -> reset -> step
    assert lib.stg_reset(ctx, None, None, None, C.c_uint64(0), ptr(obs), None) < 0
    assert lib.stg_step(ctx, ptr(act), 0, ptr(obs), ptr(rew), None, None, ptr(te), ptr(tr), None, None) < 0
    p = stg.devices.flatten_params(stg.DeviceFactory().create_device("stt_mram", stt_default_params()))
    assert lib.stg_set_params(ctx, C.byref(p), 0, None) < 0                       # no classes
    assert lib.stg_set_params(ctx, C.byref(p), 2, None) < 0                       # two classes need a class index
    assert lib.stg_set_params(ctx, C.byref(p), 1, None) == 0
    assert lib.stg_step(ctx, ptr(act), 0, ptr(obs), ptr(rew), None, None, ptr(te), ptr(tr), None, None) < 0   # no reset yet
    assert lib.stg_reset(ctx, None, None, None, C.c_uint64(0), ptr(obs), None) == 0
    assert lib.stg_step(ctx, None, 0, ptr(obs), ptr(rew), None, None, ptr(te), ptr(tr), None, None) < 0       # NULL actions
    assert lib.stg_step_many(ctx, 0, ptr(act), 0, 1, 0, ptr(obs), None, ptr(rew), None, None, ptr(te), ptr(tr), None, None) < 0
    assert lib.stg_step(ctx, ptr(act), 0, ptr(obs), ptr(rew), None, None, ptr(te), ptr(tr), None, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all()
    out = (C.c_uint64 * 4)()
    assert lib.stg_get_counters(ctx, out, 0) == 0 and out[0] == n
    assert lib.stg_get_counters(None, out, 0) < 0 and lib.stg_thermal_strength(ctx, 5, C.byref(C.c_double())) < 0
    lib.stg_destroy(ctx)
    lib.stg_destroy(None)                                                          # a no-op, like free(NULL)


@pytest.mark.parametrize("solver,thermal", [("rk4", False), ("rk4", True), ("rk45", True), ("euler", False)])
def test_per_env_parameters(stg, solver, thermal):
    """stg_set_params_per_env: every env carries its own device record.  (i) against the class-table path
    with one class per env (<= 64 envs): bit-identical, the lane derives its constants with the host's arithmetic;
    (ii) against the oracle with one class per env at 200 envs incl. tilted axes and out-of-range values (validator
    gate -> no-op); (iii) sizes that take the 4-wavefront layout elsewhere (65 536 envs) run and keep |m| = 1."""
    from helpers import OracleBackend, unit_rows
    rng = np.random.default_rng(77)
    vol0 = 9.7e-6 if solver == "rk45" else 8.75e-11

    def variation(n, tilt):
        ov = dict(volume=vol0 * 10 ** rng.uniform(-0.3, 0.3, n), damping=10 ** rng.uniform(-2.2, -1.0, n),
                  uniaxial_anisotropy=rng.uniform(6e5, 1.4e6, n), saturation_magnetization=rng.uniform(6e5, 1.0e6, n),
                  polarization=rng.uniform(0.3, 0.9, n), resistance_parallel=rng.uniform(800, 1500, n))
        if tilt:
            ax = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1)); ax[:, :2] = rng.normal(0, 0.2, (n, 2))
            ov["easy_axis"] = ax
        if solver == "rk45":
            # the fields only LLGSSolver reads (llgs_solver.py:200-209) -- the 20-double record layout of the RK45 contexts
            dm = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1)); dm[:, 2] = rng.uniform(0.8, 1.0, n)
            if tilt:
                dm[:, 0] = rng.uniform(0.0, 0.1, n); dm[:, 1] = rng.uniform(0.0, 0.1, n)
            ov["demag_factors"] = dm
            ov["exchange_constant"] = np.where(rng.integers(0, 2, n) == 0, 0.0, rng.uniform(1e-11, 3e-11, n))
        return ov

    def dicts(n, ov):
        out = []
        for i in range(n):
            d = stt_default_params()
            for k, v in ov.items():
                d[k] = np.array(v[i]) if np.ndim(v[i]) else float(v[i])
            out.append(d)
        return out

    for n, tilt, backend_ref in ((64, False, None), (200, True, OracleBackend)):
        ov = variation(n, tilt)
        if backend_ref is OracleBackend:
            ov["uniaxial_anisotropy"][3] = 500.0        # below the validator's 1e3: that env's steps are no-ops
        m0 = unit_rows(rng, n)
        tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
        acts = [_uniform_actions(2e6, 1e-10, 2e-10)(rng, n, k) for k in range(2)]
        kw = dict(include_thermal_fluctuations=thermal, solver=solver, seed=5)
        e1 = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=vol0), per_env_params=ov, **kw)
        e2 = stg.SpinTorqueVecEnv(n, diagnostics=True, device_type=["stt_mram"] * n, device_params=dicts(n, ov),
                                  class_index=np.arange(n).astype(np.uint8), backend=backend_ref, **kw)
        outs = []
        for env in (e1, e2):
            env.reset(options={"initial_state": m0, "target_state": tgt})
            rec = []
            for a in acts:
                o, r, te, tr, info = env.step(torch.from_numpy(a))
                rec.append((o.cpu().numpy().copy(), info["reward_f64"].cpu().numpy().copy(), info["status"].cpu().numpy().copy(),
                            env.get_state()["m"].cpu().numpy().copy()))
            outs.append(rec)
            env.close()
        for (o1, r1, s1, m1), (o2, r2, s2, m2) in zip(*outs):
            assert np.array_equal(s1, s2), (n, solver)
            if backend_ref is None:
                assert np.array_equal(m1, m2) and np.array_equal(o1, o2) and np.array_equal(r1, r2), (n, solver)
            else:
                tol = (TOL_RK45 if solver == "rk45" else TOL_RK4) * (50 if thermal else 1)
                assert np.abs(m1 - m2).max() <= tol, (n, solver, np.abs(m1 - m2).max())
                assert np.allclose(o1, o2, rtol=3e-7, atol=1e-9) and np.allclose(r1, r2, rtol=1e-9, atol=1e-9)
        if backend_ref is OracleBackend and solver != "rk45":      # (the gate belongs to RobustLLGSSolver, LLGSSolver has none)
            assert outs[0][0][2][3] == 1              # STG_STATUS_NOOP for the env the validator gate rejects
    if solver == "rk4" and not thermal:
        n = 65536
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=vol0), per_env_params=variation(n, False),
                                   include_thermal_fluctuations=True, solver=solver, seed=1, autoreset=True)
        env.reset(seed=0)
        a = torch.from_numpy(_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, 0))
        for _ in range(2):
            obs, *_ = env.step(a)
        m = env.get_state()["m"]
        assert torch.isfinite(obs).all() and torch.all(torch.abs(torch.linalg.norm(m, dim=0) - 1) < 1e-14)
        env.close()


@pytest.mark.parametrize("solver", ["rk4", "euler"])
def test_per_env_parameters_device_physics_mixed_types(stg, solver):
    """Per-env parameter records of a device-physics context (24-double layout: core + SOT factors / sigma + VCMA coefficients,
    csrc/stg_kernels.hpp: EnvParams) over a mixed STT / SOT / VCMA batch: bit-identical to the class-table path with one class per
    env, whose constants the host derives (same arithmetic, sot_mram.py:61-72,163-194, vcma_mram.py:122-147)."""
    from helpers import unit_rows
    rng = np.random.default_rng(12)
    n = 63
    fac = stg.DeviceFactory()
    vol = 8.75e-11
    bases = []
    for t in ("stt_mram", "sot_mram", "vcma_mram"):
        d = fac.get_default_parameters(t); d.update(polarization=0.7, volume=vol)
        bases.append(d)
    types = ["stt_mram", "sot_mram", "vcma_mram"]
    cls = rng.integers(0, 3, n).astype(np.uint8)
    ov = dict(volume=vol * 10 ** rng.uniform(-0.2, 0.2, n), damping=10 ** rng.uniform(-2.2, -1.2, n),
              uniaxial_anisotropy=rng.uniform(8e5, 1.2e6, n), polarization=rng.uniform(0.5, 0.9, n),
              resistance_parallel=rng.uniform(900, 1300, n))
    per_class = []
    for i in range(n):
        d = dict(bases[cls[i]])
        for k, v in ov.items():
            d[k] = float(v[i])
        per_class.append(d)
    m0 = unit_rows(rng, n)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    acts = [_uniform_actions(2e6, 1e-10, 3e-10)(rng, n, k) for k in range(2)]
    kw = dict(include_thermal_fluctuations=False, solver=solver, seed=5, torque_model="device")
    e1 = stg.SpinTorqueVecEnv(n, diagnostics=True, device_type=types, device_params=bases, class_index=cls, per_env_params=ov, **kw)
    e2 = stg.SpinTorqueVecEnv(n, diagnostics=True, device_type=[types[c] for c in cls], device_params=per_class,
                              class_index=np.arange(n).astype(np.uint8), **kw)
    outs = []
    for env in (e1, e2):
        env.reset(options={"initial_state": m0, "target_state": tgt})
        rec = []
        for a in acts:
            o, r, te, tr, info = env.step(torch.from_numpy(a))
            rec.append((o.clone(), info["reward_f64"].clone(), info["status"].clone(), info["energy"].clone(), env.get_state()["m"].clone()))
        outs.append(rec)
        env.close()
    for k, (x, y) in enumerate(zip(*outs)):
        for j, (p, q) in enumerate(zip(x, y)):
            assert torch.equal(p, q), (solver, k, j)
    assert float((outs[0][-1][4] - torch.from_numpy(m0.T).cuda()).abs().max()) > 1e-3       # (the pulses did move the magnetisation)


def test_switching_statistics_independent_streams(stg):
    """BASELINE config 3 gate: with the thermal field on, outcome statistics must match the CPU restatement run with its
    OWN random stream.  In every regime the validator admits, the reference's Brown field (no 1/sqrt(dt)) is
    too weak to switch anything by itself (final-state spread ~1e-6, golden G10); the one place where it decides the
    outcome is the spin-torque-dominated small-volume regime, where it tips a sub-step between "components overflow ->
    reset to +z -> solve succeeds" and "norm overflows -> zero row -> solve fails, state kept" (H3).  There the fraction
    of failed solves and the fraction of switched envs are genuinely random: HIP (65 536 envs, seed A) and oracle
    (4096 envs, seed B) must agree within the binomial error of the smaller sample."""
    from helpers import OracleBackend
    par = stt_default_params(volume=1e-28)
    m0 = np.array([0.05, 0.0, 1.0]); m0 /= np.linalg.norm(m0)
    stats = []
    for n, backend, seed in ((65536, None, 11), (4096, OracleBackend, 22)):
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=par, include_thermal_fluctuations=True, solver="rk4", seed=seed, backend=backend)
        env.reset(options={"initial_state": np.tile(m0, (n, 1)), "target_state": np.tile([0.0, 0.0, -1.0], (n, 1))})
        a = np.empty((n, 2), dtype=np.float32); a[:, 0] = 5e5; a[:, 1] = 2e-10
        _, _, te, tr, info = env.step(torch.from_numpy(a))
        st = info["status"].cpu().numpy()
        mz = env.get_state()["m"].cpu().numpy()[2]
        stats.append((float((st == 1).mean()), float((mz < 0).mean()), n))
        env.close()
    (f_hip, s_hip, _), (f_cpu, s_cpu, n_cpu) = stats
    print("failed solves: HIP %.4f CPU %.4f; switched: HIP %.4f CPU %.4f" % (f_hip, f_cpu, s_hip, s_cpu))
    assert 0.05 < f_cpu < 0.95 and 0.05 < s_cpu < 0.95            # the regime is genuinely stochastic
    for a_, b_ in ((f_hip, f_cpu), (s_hip, s_cpu)):
        sigma = np.sqrt(b_ * (1 - b_) / n_cpu)
        assert abs(a_ - b_) <= 4 * sigma + 1e-3, (a_, b_, sigma)


@pytest.mark.parametrize("n", [65537, 70001, 100000, 131072, 131072 + 4096 + 77, 300000])
def test_schedule_covers_every_env_exactly_once(stg, n):
    """Launch sizes that take the 4-wavefront workgroups with ragged tiles: the sorted, XCD-aware schedule (and the
    workgroup composition rules) must step every env exactly once -- same results as the identity schedule, bit for bit,
    and the on-device counters must add up."""
    rng = np.random.default_rng(n)
    acts = torch.from_numpy(_uniform_actions(2e6, 1e-10, 4e-10)(rng, n, 0))
    outs = []
    for ls in (False, True):
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=True,
                                   solver="rk4", seed=3, lane_sort=ls)
        env.reset(seed=7)
        o, r, te, tr, info = env.step(acts)
        st = env.get_state()
        c = env.backend.counters()
        assert c["env_steps"] == n, (n, ls, c)
        outs.append((o.clone(), info["reward_f64"].clone(), st["m"].clone(), st["step_count"].clone()))
        env.close()
    for x, y in zip(*outs):
        assert torch.equal(x, y), (n,)
    assert bool((outs[0][3] == 1).all())


def test_rk45_zero_error_norm_and_fixed_points(stg):
    """m along the easy axis with no current is a fixed point: every RK45 error norm is exactly 0 (SciPy: factor =
    MAX_FACTOR), and the hard axis is an unstable one.  The kernel's clamps must behave at err = 0 as SciPy's do."""
    from helpers import OracleBackend
    n = 64
    m0 = np.tile([0.0, 0.0, 1.0], (n, 1)); m0[1] = [0.0, 0.0, -1.0]; m0[2] = [1.0, 0.0, 0.0]; m0[3] = [0.6, 0.0, 0.8]
    tgt = np.tile([0.0, 0.0, 1.0], (n, 1))
    res = []
    for backend in (None, OracleBackend):
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(), include_thermal_fluctuations=False, solver="rk45", backend=backend)
        env.reset(options={"initial_state": m0, "target_state": tgt})
        a = np.zeros((n, 2), dtype=np.float32); a[:, 1] = 2e-11
        o, r, te, tr, info = env.step(torch.from_numpy(a))
        res.append((env.get_state()["m"].cpu().numpy().copy(), info["status"].cpu().numpy().copy()))
        env.close()
    assert np.array_equal(res[0][1], res[1][1]) and np.abs(res[0][0] - res[1][0]).max() <= 1e-12
    assert (res[0][1] == 0).all() and np.array_equal(res[0][0][:, 0], [0.0, 0.0, 1.0])


@pytest.mark.parametrize("solver", ["euler", "rk4", "rk45"])
def test_results_do_not_depend_on_wavefront_composition(stg, solver):
    """The tile sort ranks the envs of one duration bucket in the order their LDS atomics arrive, so which envs share a
    wavefront varies from run to run.  A lane's arithmetic must not depend on its wavefront-mates (the wave-uniform fast
    paths may only skip work): repeated runs of the same step -- many envs per bucket, short and long pulses mixed so
    that the fast and the general normalisation meet in one wavefront -- must agree bit for bit, thermal on and off."""
    n = 8192
    rng = np.random.default_rng(5)
    vol = 9.7e-6 if solver == "rk45" else 8.75e-11
    a = np.empty((n, 2), dtype=np.float32)
    a[:, 0] = rng.choice([0.0, 1e6, -2e6], n)
    a[:, 1] = rng.choice([1e-12, 2e-11, 1.5e-10], n)             # three buckets, thousands of envs each
    for thermal in (False, True):
        ref = None
        for rep in range(5):
            env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=vol), include_thermal_fluctuations=thermal,
                                       solver=solver, seed=3)
            env.reset(seed=9)
            o, r, te, tr, info = env.step(torch.from_numpy(a))
            cur = (env.get_state()["m"].clone(), info["reward_f64"].clone(), o.clone())
            env.close()
            if ref is None:
                ref = cur
            else:
                assert all(torch.equal(x, y) for x, y in zip(ref, cur)), (solver, thermal, rep)
    if solver == "rk45":
        return
    # strong-noise regime: |m|^2 after a sub-step scatters around the threshold of the fast normalisation, so most
    # wavefronts mix both paths whatever the sort does
    n = 4096
    a = np.zeros((n, 2), dtype=np.float32)
    a[:, 1] = rng.uniform(1e-10, 3e-10, n).astype(np.float32)
    for noise in ("white", "ou"):
        ref = None
        for rep in range(6):
            env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=1e-27), include_thermal_fluctuations=True,
                                       solver=solver, seed=13, noise_model=noise, correlation_time=3e-12)
            env.reset(seed=9)
            env.step(torch.from_numpy(a))
            cur = env.get_state()["m"].clone()
            env.close()
            if ref is None:
                ref = cur
            else:
                assert torch.equal(ref, cur), (solver, noise, rep, int((ref != cur).any(dim=0).sum()))


@pytest.mark.parametrize("snake,walk", [("1", None), ("1", "2"), ("0", "3"), (None, "1")])
def test_schedule_knobs_keep_the_slot_map_a_bijection(stg, monkeypatch, snake, walk):
    """The experiment knobs of the sorted schedule (STG_SNAKE: boustrophedon rounds forced on/off; STG_WALK_TILES: tiles an
    XCD group walks together) only permute which wavefront integrates which 64 slots: for every setting and ragged launch
    size each env is stepped exactly once, with the bits of the identity schedule."""
    for key, val in (("STG_SNAKE", snake), ("STG_WALK_TILES", walk)):
        if val is None:
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, val)
    for n in (100000, 131072 + 8192, 300000, 589824 + 4096 * 3 + 5):
        rng = np.random.default_rng(n)
        acts = torch.from_numpy(_uniform_actions(2e6, 1e-10, 2.5e-10)(rng, n, 0))
        outs = []
        for ls in (False, True):
            env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=False,
                                       solver="rk4", seed=3, lane_sort=ls)
            env.reset(seed=7)
            o, r, te, tr, info = env.step(acts)
            st = env.get_state()
            assert env.backend.counters()["env_steps"] == n, (n, ls)
            outs.append((o.clone(), info["reward_f64"].clone(), st["m"].clone(), st["step_count"].clone()))
            env.close()
        for x, y in zip(*outs):
            assert torch.equal(x, y), (n, snake, walk)
        assert bool((outs[0][3] == 1).all())


def test_placement_table_of_a_step_launch(stg):
    """ABI v4, stg_get_placement: every wavefront of a step launch records the SIMD it ran on (the schedules lean
    on observed dispatcher behaviour; the shipped library can report what a launch actually got).  65 536 envs, RK45 + thermal = 1024
    workgroups of one integrating + one producing wavefront; 4096 envs at T = 0 K = 64 one-wavefront workgroups; the ring keeps the
    last 32 launches."""
    n = 65536
    rng = np.random.default_rng(0)
    env = stg.SpinTorqueVecEnv(n, device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=True, solver="rk45", seed=2,
                               autoreset=True)
    env.reset(seed=1)
    with pytest.raises(Exception, match="launches_back"):
        env.backend.placement(0)                                    # nothing launched yet
    for k in range(3):
        env.step(torch.from_numpy(_uniform_actions(2e6, 1e-10, 2e-10)(rng, n, k)))
    for back in range(3):
        p = env.backend.placement(back)
        assert p["workgroups"] == 1024 and p["waves_per_workgroup"] == 2 and p["waves_recorded"] == 2048
        assert sum(k * v for k, v in p["integrating_per_simd"].items()) == 1024           # 1024 integrating + 1024 producing wavefronts
        assert 512 <= p["simds_used"] <= 1024 and 0 <= p["simd_double_booked"] <= 512
    with pytest.raises(Exception, match="launches_back"):
        env.backend.placement(3)
    print("placement of the 65 536-env wave-specialised launch:", p)
    env.close()
    env = stg.SpinTorqueVecEnv(4096, device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=False, solver="rk45", seed=2)
    env.reset(seed=1)
    for k in range(40):                                             # more launches than the ring holds
        env.step(torch.from_numpy(_uniform_actions(2e6, 1e-11, 2e-11)(rng, 4096, k)))
    p = env.backend.placement(31)
    assert p["workgroups"] == 64 and p["waves_per_workgroup"] == 1 and p["waves_recorded"] == 64 and p["integrating_per_simd"] == {1: 64}
    with pytest.raises(Exception, match="launches_back"):
        env.backend.placement(32)
    env.close()
