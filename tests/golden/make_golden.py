#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED Python reference.

Runs only in the build container (the reference does not travel to the GPU box).  The reference
package at /root/reference is imported as-is through a stand-in for the missing `gymnasium`
dependency (tests/golden/gym_stub.py); nothing of the reference is copied -- the .npz files hold
inputs and the outputs the reference computed for them.

Protocol: result/observation caches neutralised (H1/H2), states injected
through reset(options=...), every LLGSSolver.solve under a SIGALRM guard, solver wall-clock
timeout lifted (it only guards; 5 ns pulses take ~1.1 s on this machine, close to the 2 s limit).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [G1 G2 ...]
"""
import logging
import os
import signal
import sys
import time
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gym_stub  # noqa: E402

gym_stub.install()
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")
logging.disable(logging.CRITICAL)

import numpy as np  # noqa: E402

from spin_torque_gym.devices import DeviceFactory  # noqa: E402
from spin_torque_gym.envs.spin_torque_env import SpinTorqueEnv  # noqa: E402
from spin_torque_gym.physics.llgs_solver import LLGSSolver  # noqa: E402
from spin_torque_gym.physics.simple_solver import SimpleLLGSSolver  # noqa: E402
from spin_torque_gym.physics.thermal_model import ThermalFluctuations  # noqa: E402
from spin_torque_gym.utils.performance import get_optimizer  # noqa: E402
from spin_torque_gym.utils.robust_solver import RobustLLGSSolver  # noqa: E402


class Timeout(Exception):
    pass


def _alarm(signum, frame):
    raise Timeout()


signal.signal(signal.SIGALRM, _alarm)


def guarded(seconds, fn, *a, **k):
    signal.alarm(seconds)
    try:
        return fn(*a, **k)
    finally:
        signal.alarm(0)


def stt_params(**over):
    p = DeviceFactory().get_default_parameters("stt_mram")
    p.update(over)
    return p


def unit_rows(rng, n):
    v = rng.normal(0, 1, (n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def robust_solver():
    # the env's constructor arguments (envs/spin_torque_env.py:93-102), wall-clock guard lifted
    return RobustLLGSSolver(method="rk4", rtol=1e-3, atol=1e-6, timeout=1e9, max_retries=2,
                            fallback_method="euler", enable_monitoring=True, enable_validation=True)


def pulse(J, T):
    return lambda t: J if t <= T else 0.0


def zero_field(t):
    return np.zeros(3)


def run_robust(solver, m0, T, params, J, thermal=False, temperature=300.0):
    get_optimizer().cache.clear()
    return solver.solve(m0.copy(), (0, T), params, pulse(J, T), zero_field, thermal, temperature)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path} ({os.path.getsize(path)} bytes)")


# ------------------------------------------------------------------------------------------------
def G1():
    """simple_rk4_relax: default STT params, J=0."""
    rng = np.random.default_rng(101)
    solver = robust_solver()
    params = stt_params()
    m0s = np.concatenate([unit_rows(rng, 29), np.array([[0.3, 0.2, 0.93]]) / np.linalg.norm([0.3, 0.2, 0.93]),
                          [[0.0, 0.0, 1.0]], [[1.0, 0.0, 0.0]]])
    Ts = np.array([1e-10, 1e-9, 5e-9, float(np.float32(1e-9)), float(np.float32(3.3e-10)), 1e-12, 5e-12])
    rows = []
    for T in Ts:
        sel = range(len(m0s)) if T < 2e-9 else range(8)
        for i in sel:
            r = run_robust(solver, m0s[i], T, params, 0.0)
            rows.append((i, T, r["success"], *r["m"][-1], r.get("n_steps", -1)))
    rows = np.array(rows, dtype=float)
    # two full trajectories
    r = run_robust(solver, m0s[0], 1e-9, params, 0.0)
    r2 = run_robust(solver, m0s[29], float(np.float32(3.3e-10)), params, 0.0)
    save("G1_simple_rk4_relax", m0=m0s, m0_index=rows[:, 0].astype(int), T=rows[:, 1], success=rows[:, 2].astype(bool),
         m_final=rows[:, 3:6], n_steps=rows[:, 6].astype(int), traj0_m=r["m"], traj0_t=r["t"],
         traj1_m=r2["m"], traj1_t=r2["t"])


def G2():
    """simple_rk4_stt: rescaled volume so the STT term is well conditioned; float32 durations (H4/H5)."""
    rng = np.random.default_rng(202)
    solver = robust_solver()
    m0s = np.concatenate([unit_rows(rng, 6), [[0.0, 0.0, 1.0]], [[0.02, -0.01, 0.9997]], [[0.01, 0.02, -0.9997]]])
    m0s = m0s / np.linalg.norm(m0s, axis=1, keepdims=True)
    durations = [float(np.float32(x)) for x in (1e-10, 2.5e-10, 7.7e-10, 1e-9, 1.3333e-9, 2e-9)] + [1e-9, 5e-10]
    rows = []
    for vol in (8.75e-11, 2e-11):
        params = stt_params(volume=vol)
        for J in (2e6, -2e6, 5e5):
            for T in durations:
                for i in range(len(m0s)):
                    if (i + int(T * 1e13)) % 3:       # thin the grid
                        continue
                    r = run_robust(solver, m0s[i], T, params, J)
                    rows.append((vol, J, T, i, r["success"], *r["m"][-1], r.get("n_steps", -1)))
    rows = np.array(rows, dtype=float)
    params = stt_params(volume=8.75e-11)
    r = run_robust(solver, m0s[7], 1e-9, params, 2e6)
    save("G2_simple_rk4_stt", m0=m0s, volume=rows[:, 0], J=rows[:, 1], T=rows[:, 2], m0_index=rows[:, 3].astype(int),
         success=rows[:, 4].astype(bool), m_final=rows[:, 5:8], n_steps=rows[:, 8].astype(int),
         traj_m=r["m"], traj_t=r["t"], traj_volume=8.75e-11, traj_J=2e6, traj_T=1e-9, traj_m0_index=7)


def G3():
    """simple_degenerate: default params, J != 0 -> overflow semantics."""
    rng = np.random.default_rng(303)
    solver = robust_solver()
    params = stt_params()
    m0s = np.concatenate([unit_rows(rng, 5), [[0.0, 0.0, 1.0]]])
    rows = []
    for J in (1.0, 1e2, 1e3, 2e3, 2.5e3, 3e3, 4e3, 1e4, 1e5, 1e6, -1e6, 2e6, 1e8):
        for T in (1e-10, 1e-9):
            for i in range(len(m0s)):
                r = run_robust(solver, m0s[i], T, params, J)
                # plain SimpleLLGSSolver for the raw trajectory (which row first became zero / reset)
                get_optimizer().cache.clear()
                s = SimpleLLGSSolver(method="rk4", rtol=1e-3, atol=1e-6, timeout=1e9)
                raw = s.solve(m0s[i].copy(), (0, T), params, pulse(J, T), zero_field, False, 300.0)
                norms = np.linalg.norm(raw["m"], axis=1)
                zero_rows = np.nonzero(norms < 1e-12)[0]
                first_zero = int(zero_rows[0]) if len(zero_rows) else -1
                rows.append((J, T, i, r["success"], *r["m"][-1], first_zero, *raw["m"][-1]))
    rows = np.array(rows, dtype=float)
    save("G3_simple_degenerate", m0=m0s, J=rows[:, 0], T=rows[:, 1], m0_index=rows[:, 2].astype(int),
         success=rows[:, 3].astype(bool), robust_m_last=rows[:, 4:7], first_zero_row=rows[:, 7].astype(int),
         raw_m_last=rows[:, 8:11])


def _llgs_cases(name, cases, params_of):
    solver = LLGSSolver()        # RK45, rtol 1e-6, atol 1e-9, max_step 1e-12, gamma 2.21e5
    out = {}
    meta = []
    for k, (m0, T, J, tag) in enumerate(cases):
        params = params_of(tag)
        t0 = time.time()
        r = guarded(120, solver.solve, np.array(m0, dtype=float), (0, T), params, pulse(J, T), zero_field,
                    thermal_noise=False, temperature=300.0)
        print(f"    {name} case {k}: {len(r['t'])} points in {time.time() - t0:.2f}s")
        out[f"t_{k}"] = r["t"]
        out[f"m_{k}"] = r["m"]
        out[f"energy_{k}"] = r["energy"]
        out[f"torques_{k}"] = r["torques"]
        meta.append((*m0, T, J, tag, bool(r["success"])))
    out["cases"] = np.array(meta, dtype=float)
    save(name, **out)


def G4():
    """llgs_rk45_relax: default STT params, J=0, SciPy RK45."""
    rng = np.random.default_rng(404)
    m0s = np.concatenate([[np.array([0.3, 0.2, 0.93]) / np.linalg.norm([0.3, 0.2, 0.93])], unit_rows(rng, 3)])
    cases = [(tuple(m), T, 0.0, 0) for m in m0s for T in (1e-10, 1e-9)]
    cases.append((tuple(m0s[1]), float(np.float32(4.2e-10)), 0.0, 0))
    _llgs_cases("G4_llgs_rk45_relax", cases, lambda tag: stt_params())


def G5():
    """llgs_rk45_stt: volume rescaled so the Slonczewski term is well conditioned for RK45."""
    vols = {0: 9.7e-6, 1: 2e-6}
    m_up = np.array([0.02, -0.01, 0.9997])
    m_up /= np.linalg.norm(m_up)
    m_dn = np.array([0.01, 0.02, -0.9997])
    m_dn /= np.linalg.norm(m_dn)
    m_r = np.array([0.5, -0.6, 0.3])
    m_r /= np.linalg.norm(m_r)
    cases = [(tuple(m_up), 1e-9, 2e6, 0), (tuple(m_dn), 1e-9, -2e6, 0), (tuple(m_up), 1e-9, 2e6, 1),
             (tuple(m_r), float(np.float32(6e-10)), -2e6, 1), (tuple(m_r), 5e-10, 5e5, 0)]
    _llgs_cases("G5_llgs_rk45_stt", cases, lambda tag: stt_params(volume=vols[int(tag)]))


def _episode(env, m0, target, actions):
    env.cache_observations = False          # H2
    env.solver.timeout = 1e9                # wall-clock guard only
    obs0, _ = env.reset(seed=0, options={"initial_state": np.array(m0, dtype=float),
                                        "target_state": np.array(target, dtype=float)})
    rec = dict(obs=[obs0], reward=[], terminated=[], truncated=[], energy=[], success=[], m=[env.current_magnetization.copy()],
               total_energy=[])
    for a in actions:
        get_optimizer().cache.clear()       # H1
        o, r, te, tr, info = env.step(np.array(a, dtype=np.float32))
        rec["obs"].append(o)
        rec["reward"].append(r)
        rec["terminated"].append(te)
        rec["truncated"].append(tr)
        rec["energy"].append(info.get("energy_consumed", np.nan))
        rec["success"].append(info.get("simulation_success", False))
        rec["m"].append(env.current_magnetization.copy())
        rec["total_energy"].append(env.total_energy)
    return {k: np.array(v) for k, v in rec.items()}


def G6():
    """env_episode: SpinTorqueEnv (continuous/vector, thermal off), scripted actions."""
    out = {}
    episodes = []

    def add(tag, env_kwargs, m0, target, actions, dev_type="stt_mram", params=None):
        env = SpinTorqueEnv(device_type=dev_type, device_params=params, include_thermal_fluctuations=False, **env_kwargs)
        rec = _episode(env, m0, target, actions)
        k = len(episodes)
        for name, arr in rec.items():
            out[f"ep{k}_{name}"] = arr
        out[f"ep{k}_actions"] = np.array(actions, dtype=np.float32)
        out[f"ep{k}_m0"] = np.array(m0, dtype=float)
        out[f"ep{k}_target"] = np.array(target, dtype=float)
        episodes.append(tag)
        print(f"    G6 episode {k} ({tag}): {len(actions)} steps")

    m0 = np.array([0.3, 0.2, 0.93]) / np.linalg.norm([0.3, 0.2, 0.93])
    # 0: default params, J=0 relaxation, varied float32 durations
    add("default_relax", {}, m0, [0, 0, -1],
        [(0.0, 1e-9), (0.0, 3.3e-10), (0.0, 5e-12), (0.0, 2e-9), (0.0, 0.0), (0.0, 7.7e-10)])
    # 1: default params, J != 0: solver fails (H3), m unchanged, energy charged, reward bonus (H7)
    add("default_noop", {}, m0, [0, 0, -1],
        [(1e6, 1e-9), (-2e6, 5e-10), (5e6, 1e-9), (2e3, 1e-10), (0.0, 1e-10)])
    # 2: well-conditioned volume: switching +z -> -z with J>0, terminates on success
    p2 = stt_params(volume=8.75e-11)
    add("stt_switch", {}, [0.02, -0.01, 0.9997], [0, 0, -1],
        [(2e6, 1e-9), (2e6, 1e-9), (2e6, 1e-9), (1e6, 5e-10)], params=p2)
    # 3: same params, bad / out-of-range actions, small max_steps -> truncation
    add("stt_bad_actions", {"max_steps": 5}, [0.6, 0.0, 0.8], [0, 0, 1],
        [(np.nan, 1e-9), (1e6, np.inf), (-9e9, 1e-3), (1e6, -1.0), (5e5, 2e-10), (5e5, 2e-10)], params=p2)
    # 4: non-default thresholds / weights / max_current / max_duration
    add("stt_custom_cfg", {"max_current": 1e6, "max_duration": 1e-9, "success_threshold": 0.5,
                           "energy_penalty_weight": 0.3, "temperature": 350.0, "max_steps": 20},
        [0.1, 0.3, -0.9], [0, 0, 1], [(-2e6, 2e-9), (-1e6, 1e-9), (-1e6, 1e-9), (-1e6, 1e-9)], params=p2)
    # 5: SOT with the factory defaults (+easy_axis already present, no 'polarization'): validation fails -> no-op
    fac = DeviceFactory()
    add("sot_default_noop", {}, m0, [0, 0, -1], [(1e6, 1e-9), (0.0, 1e-10)], dev_type="sot_mram",
        params=fac.get_default_parameters("sot_mram"))
    # 6: SOT with polarization and rescaled volume
    p6 = fac.get_default_parameters("sot_mram")
    p6.update(polarization=0.7, volume=8.75e-11)
    add("sot_polarized", {}, [0.02, -0.01, 0.9997], [0, 0, -1], [(2e6, 1e-9), (2e6, 5e-10), (0.0, 1e-10)],
        dev_type="sot_mram", params=p6)
    # 7: VCMA with polarization and rescaled volume, tilted easy axis and reference layer
    p7 = fac.get_default_parameters("vcma_mram")
    p7.update(polarization=0.6, volume=5e-11, easy_axis=np.array([0.1, 0.0, 1.0]),
              reference_magnetization=np.array([0.0, 0.2, 1.0]))
    add("vcma_polarized_tilted", {}, [0.3, 0.1, 0.9], [0, 0, -1], [(2e6, 1e-9), (-2e6, 5e-10), (1e6, 3e-10)],
        dev_type="vcma_mram", params=p7)
    # 8: temperature = 0 -> RobustLLGSSolver input validation rejects -> every step is a no-op
    add("temperature_zero_noop", {"temperature": 0.0}, m0, [0, 0, 1], [(0.0, 1e-9), (1e6, 1e-9)])
    out["episode_tags"] = np.array(episodes)
    save("G6_env_episode", **out)


def G7():
    """resistance: compute_resistance over a sphere grid for the three device classes."""
    rng = np.random.default_rng(707)
    ms = np.concatenate([unit_rows(rng, 40), [[0, 0, 1.0]], [[0, 0, -1.0]], [[1.0, 0, 0]]])
    fac = DeviceFactory()
    out = {"m": ms}
    for t in ("stt_mram", "sot_mram", "vcma_mram"):
        dev = fac.create_device(t, fac.get_default_parameters(t))
        out[f"R_{t}"] = np.array([dev.compute_resistance(m.copy()) for m in ms])
        p = fac.get_default_parameters(t)
        p.update(reference_magnetization=np.array([0.0, 0.2, 1.0]), resistance_parallel=1234.5,
                 resistance_antiparallel=3210.0)
        dev = fac.create_device(t, p)
        out[f"R_{t}_tilted"] = np.array([dev.compute_resistance(m.copy()) for m in ms])
    # non-unit input (STT renormalises, SOT/VCMA do not)
    ms2 = ms * 1.7
    for t in ("stt_mram", "sot_mram", "vcma_mram"):
        dev = fac.create_device(t, fac.get_default_parameters(t))
        out[f"R_{t}_scaled"] = np.array([dev.compute_resistance(m.copy()) for m in ms2])
    save("G7_resistance", **out)


def G8():
    """thermal strengths (both Boltzmann constants) and white-noise moments."""
    grid = []
    simple = SimpleLLGSSolver(method="rk4")
    for alpha in (0.005, 0.01, 0.05):
        for ms_ in (6e5, 8e5, 1.2e6):
            for vol in (1e-24, 1e-23, 8.75e-11):
                for T in (77.0, 300.0, 400.0):
                    tf = ThermalFluctuations(temperature=T)
                    s_llgs = tf.compute_noise_strength(alpha, ms_, vol)
                    # SimpleLLGSSolver strength: difference of two effective fields is not observable
                    # (random), so evaluate its expression through the solver's own constants
                    kb = 1.38e-23
                    s_simple = np.sqrt(2 * alpha * kb * T / (simple.mu_0 * ms_ * vol * simple.gamma))
                    grid.append((alpha, ms_, vol, T, s_llgs, s_simple))
    grid = np.array(grid)
    # empirical check of the SimpleLLGSSolver field: std of h_eff over many draws at m = z
    params = stt_params()
    np.random.seed(12345)
    draws = np.array([simple._compute_effective_field(np.array([0.0, 0.0, 1.0]), params, None, 0.0, 300.0)
                      for _ in range(4000)])
    base = simple._compute_effective_field(np.array([0.0, 0.0, 1.0]), params, None, 0.0, 0.0)
    tf = ThermalFluctuations(temperature=300.0, seed=7)
    white = np.array([tf.generate_thermal_field(0.01, 800e3, params["volume"], 1e-12, correlated=False)
                      for _ in range(4000)])
    save("G8_thermal", grid=grid, simple_field_std=(draws - base).std(axis=0), simple_field_mean=(draws - base).mean(axis=0),
         tf_white_std=white.std(axis=0), tf_white_mean=white.mean(axis=0),
         default_strength_llgs=ThermalFluctuations(300.0).compute_noise_strength(0.01, 800e3, params["volume"]))


def G9():
    """reset_seeds: reset(seed=s) -> (m0, target, obs)."""
    env = SpinTorqueEnv(include_thermal_fluctuations=False)
    env.cache_observations = False
    rows = []
    obs = []
    for s in range(64):
        o, _ = env.reset(seed=s)
        rows.append((*env.current_magnetization, *env.target_magnetization))
        obs.append(o)
    # consecutive resets without a seed continue the stream of the last seed
    env.reset(seed=1234)
    cont = []
    for _ in range(4):
        env.reset()
        cont.append((*env.current_magnetization, *env.target_magnetization))
    save("G9_reset_seeds", state=np.array(rows), obs=np.array(obs), continued_from_1234=np.array(cont))


def G10():
    """thermal_diffusion: thermal field ON.  (a) regimes where the solver is well conditioned: the Brown field is
    ~1e-15 of H_k, so the reference's thermal-on result equals its thermal-off result to ~1e-10 whatever the random
    stream; (b) a volume of 1e-30 m^3 (the validator's minimum) makes the field ~1e-5 of H_k: the final state diffuses
    measurably around the deterministic trajectory -- samples of that diffusion for the statistical parity test."""
    out = {}
    solver = robust_solver()
    m0 = np.array([0.3, 0.2, 0.93]) / np.linalg.norm([0.3, 0.2, 0.93])
    rows = []
    np.random.seed(2024)
    for vol, J, T in ((50e-9 * 100e-9 * 2e-9, 0.0, 5e-10), (8.75e-11, 2e6, 1e-9), (8.75e-11, -2e6, 5e-10), (2e-11, 5e5, 1e-9)):
        params = stt_params(volume=vol)
        r_off = run_robust(solver, m0, T, params, J, thermal=False)
        r_on = run_robust(solver, m0, T, params, J, thermal=True, temperature=300.0)
        rows.append((vol, J, T, *r_off["m"][-1], *r_on["m"][-1], r_on["success"]))
    out["wellcond"] = np.array(rows, dtype=float)
    # (b) diffusion samples, SimpleLLGSSolver rk4
    params = stt_params(volume=1e-30)
    T = 2e-10
    det = run_robust(solver, m0, T, params, 0.0, thermal=False)["m"][-1]
    np.random.seed(7)
    samples = np.array([run_robust(solver, m0, T, params, 0.0, thermal=True, temperature=300.0)["m"][-1] for _ in range(4000)])
    out.update(rk4_m0=m0, rk4_T=T, rk4_volume=1e-30, rk4_deterministic=det, rk4_samples=samples)
    print(f"    rk4 diffusion: rms deviation {np.sqrt(((samples - det) ** 2).sum(axis=1).mean()):.3e}")
    # (c) diffusion samples, LLGSSolver RK45
    llgs = LLGSSolver()
    T = 1e-10
    det = guarded(60, llgs.solve, m0.copy(), (0, T), params, pulse(0.0, T), zero_field, thermal_noise=False)["m"][-1]
    np.random.seed(11)
    samples, npts = [], []
    for _ in range(1000):
        r = guarded(120, llgs.solve, m0.copy(), (0, T), params, pulse(0.0, T), zero_field, thermal_noise=True, temperature=300.0)
        samples.append(r["m"][-1])
        npts.append(len(r["t"]))
    samples = np.array(samples)
    out.update(rk45_m0=m0, rk45_T=T, rk45_volume=1e-30, rk45_deterministic=det, rk45_samples=samples, rk45_npts=np.array(npts))
    print(f"    rk45 diffusion: rms deviation {np.sqrt(((samples - det) ** 2).sum(axis=1).mean()):.3e}, points {np.mean(npts):.1f}")
    save("G10_thermal_diffusion", **out)


def G11():
    """simple_euler: SimpleLLGSSolver(method='euler') -- RobustLLGSSolver's fallback integrator -- behind the same gates."""
    rng = np.random.default_rng(1111)
    solver = RobustLLGSSolver(method="euler", rtol=1e-3, atol=1e-6, timeout=1e9, max_retries=2, fallback_method="euler",
                              enable_monitoring=True, enable_validation=True)
    m0s = unit_rows(rng, 6)
    rows = []
    for vol in (50e-9 * 100e-9 * 2e-9, 8.75e-11):
        params = stt_params(volume=vol)
        for J in ((0.0,) if vol < 1e-20 else (0.0, 2e6, -5e5)):
            for T in (1e-10, float(np.float32(2.5e-10)), 5e-10, 5e-12):
                for i in range(len(m0s)):
                    r = run_robust(solver, m0s[i], T, params, J)
                    rows.append((vol, J, T, i, r["success"], *r["m"][-1], r.get("n_steps", -1)))
    rows = np.array(rows, dtype=float)
    save("G11_simple_euler", m0=m0s, volume=rows[:, 0], J=rows[:, 1], T=rows[:, 2], m0_index=rows[:, 3].astype(int),
         success=rows[:, 4].astype(bool), m_final=rows[:, 5:8], n_steps=rows[:, 8].astype(int))


def G12():
    """device_terms: SOTMRAMDevice.compute_spin_torque and VCMAMRAMDevice._compute_effective_anisotropy (the device-class
    formulas the env never calls; they pin the opt-in device-physics torque extension)."""
    rng = np.random.default_rng(1212)
    fac = DeviceFactory()
    ms = unit_rows(rng, 48)
    Js = rng.uniform(-2e6, 2e6, 48)
    out = {"m": ms, "J": Js}
    for tag, over in (("default", {}), ("custom", dict(spin_hall_angle=0.3, heavy_metal_thickness=4e-9, interface_transparency=0.6,
                                                        field_like_efficiency=0.15, damping_like_efficiency=0.25, thickness=1.2e-9))):
        p = fac.get_default_parameters("sot_mram")
        p.update(over)
        dev = fac.create_device("sot_mram", p)
        dl, fl = zip(*[dev.compute_spin_torque(float(J), m.copy()) for m, J in zip(ms, Js)])
        out[f"sot_{tag}_tau_dl"] = np.array(dl)
        out[f"sot_{tag}_tau_fl"] = np.array(fl)
        out[f"sot_{tag}_factors"] = np.array([dev.tau_dl_factor, dev.tau_fl_factor])
    d2 = fac.create_device("sot_mram", fac.get_default_parameters("sot_mram"))
    dirn = np.array([1.0, 1.0, 0.0])
    dl, fl = zip(*[d2.compute_spin_torque(float(J), m.copy(), dirn) for m, J in zip(ms, Js)])
    out["sot_dir110_tau_dl"], out["sot_dir110_tau_fl"] = np.array(dl), np.array(fl)
    volts = np.concatenate([np.linspace(-3, 3, 25), [0.0, 1e-3, 2.0, -2.0, 10.0]])
    out["volts"] = volts
    for tag, over in (("default", {}), ("custom", dict(vcma_coefficient=60e-6, dielectric_thickness=1.4e-9, breakdown_voltage=1.5,
                                                        uniaxial_anisotropy=0.9e6))):
        p = fac.get_default_parameters("vcma_mram")
        p.update(over)
        dev = fac.create_device("vcma_mram", p)
        out[f"vcma_{tag}_keff"] = np.array([dev._compute_effective_anisotropy(float(v)) for v in volts])
    save("G12_device_terms", **out)


def G13():
    """array_env: SpinTorqueArray-v0 episodes (envs/array_env.py) for the four action modes, three coupling types,
    both observation modes, STT and SOT/VCMA devices; plus device.compute_effective_field samples."""
    from spin_torque_gym.envs.array_env import SpinTorqueArrayEnv
    rng = np.random.default_rng(1313)
    fac = DeviceFactory()
    out = {}
    tags = []

    def add(tag, kwargs, actions, seed):
        env = SpinTorqueArrayEnv(**kwargs)
        obs0, _ = env.reset(seed=seed)
        k = len(tags)
        rec = dict(obs=[obs0.reshape(-1)], reward=[], terminated=[], truncated=[], energy=[], similarity=[],
                   pattern=[env.current_pattern.copy()])
        for a in actions:
            o, r, te, tr, info = env.step(np.array(a, dtype=np.float32))
            rec["obs"].append(np.asarray(o).reshape(-1)); rec["reward"].append(r); rec["terminated"].append(te)
            rec["truncated"].append(tr); rec["energy"].append(info["energy_consumed"])
            rec["similarity"].append(info["pattern_similarity"]); rec["pattern"].append(env.current_pattern.copy())
        for name, arr in rec.items():
            out[f"ep{k}_{name}"] = np.array(arr)
        out[f"ep{k}_actions"] = np.array(actions, dtype=np.float32)
        out[f"ep{k}_target"] = env.target_pattern.copy()
        if getattr(env, "include_coupling", False):
            out[f"ep{k}_coupling"] = env.coupling_matrix.copy()
        tags.append(tag)
        print(f"    G13 episode {k} ({tag}): {len(actions)} steps, final similarity {rec['similarity'][-1]:.4f}")

    def acts(n, idx_hi, jmax=2e6):
        return [(rng.uniform(-0.5, idx_hi + 0.5), rng.uniform(-jmax, jmax), rng.uniform(1e-10, 2e-9)) for _ in range(n)]

    add("individual_dipolar", dict(array_size=(4, 4), action_mode="individual"), acts(8, 15) + [(3.0, 0.0, 1e-9)], 0)
    add("row_exchange", dict(array_size=(4, 4), action_mode="row", coupling_type="exchange", coupling_strength=0.3), acts(5, 3), 1)
    add("column_stray", dict(array_size=(3, 5), action_mode="column", coupling_type="stray_field", observation_mode="vector",
                             max_steps=4), acts(5, 4), 2)
    add("global", dict(array_size=(4, 4), action_mode="global"), [(rng.uniform(-2e6, 2e6), rng.uniform(1e-10, 2e-9)) for _ in range(3)], 3)
    add("nocoupling_custom", dict(array_size=(2, 3), action_mode="individual", include_coupling=False, max_current=1e6,
                                  max_duration=1e-9, success_threshold=0.2, energy_penalty_weight=0.3,
                                  observation_mode="vector", temperature=350.0), acts(6, 5, 3e6), 4)
    sot = fac.get_default_parameters("sot_mram"); sot.update(aspect_ratio=2.0)
    add("sot_devices", dict(array_size=(3, 3), action_mode="row", device_type="sot_mram", device_params=sot), acts(4, 2, 1e3), 5)
    vc = fac.get_default_parameters("vcma_mram"); vc.update(aspect_ratio=0.5, reference_magnetization=np.array([0.0, 0.2, 1.0]))
    add("vcma_devices", dict(array_size=(2, 2), action_mode="individual", device_type="vcma_mram", device_params=vc), acts(4, 3, 1e3), 6)
    out["episode_tags"] = np.array(tags)
    # device.compute_effective_field(m, 0) samples
    ms_ = unit_rows(rng, 12) * rng.uniform(0.5, 1.5, (12, 1))
    out["field_m"] = ms_
    for t, params in (("stt_mram", fac.get_default_parameters("stt_mram")), ("sot_mram", sot), ("vcma_mram", vc)):
        dev = fac.create_device(t, params)
        out[f"field_{t}"] = np.array([dev.compute_effective_field(m.copy(), np.zeros(3)) for m in ms_])
    save("G13_array_env", **out)


def G14():
    """thermal_ou: ThermalFluctuations.generate_thermal_field(correlated=True) sequences (thermal_model.py:113-137)."""
    out = {}
    cases = []
    for seed, tau, dt in ((7, 1e-12, 1e-12), (11, 5e-12, 1e-12), (3, 1e-12, 2.5e-13), (5, 2e-13, 1e-12)):
        tf = ThermalFluctuations(temperature=300.0, correlation_time=tau, seed=seed)
        seq = np.array([tf.generate_thermal_field(0.01, 800e3, 1e-24, dt, correlated=True) for _ in range(64)])
        out[f"field_{len(cases)}"] = seq
        cases.append((seed, tau, dt, tf.compute_noise_strength(0.01, 800e3, 1e-24)))
    out["cases"] = np.array(cases)
    save("G14_thermal_ou", **out)


def G15():
    """thermal_scalars: Neel-Brown closed forms of ThermalFluctuations (thermal_model.py:139-336)."""
    out = {}
    params = stt_params()
    small = dict(params); small["volume"] = 2e-26
    for tag, T, dp in (("a", 300.0, params), ("b", 350.0, small), ("c", 77.0, small)):
        tf = ThermalFluctuations(temperature=T, seed=5)
        st = tf.analyze_thermal_stability(dp, time_scale=10.0)
        out[f"stab_{tag}"] = np.array([st["thermal_stability_factor"], st["energy_barrier_J"], st["energy_barrier_kT"],
                                       st["switching_probability"], st["retention_time_years"], float(st["is_thermally_stable"]),
                                       st["temperature_K"]])
        e_b = dp["uniaxial_anisotropy"] * dp["volume"]
        out[f"retention_{tag}"] = np.array([tf.compute_retention_time(e_b), tf.compute_retention_time(e_b, failure_rate=1e-6, attempt_frequency=2e9)])
        out[f"times_{tag}"] = np.array([tf.sample_switching_time(e_b) for _ in range(8)])
        sw = tf.generate_temperature_sweep((50.0, 400.0), dp, n_points=9)
        for k, v in sw.items():
            out[f"sweep_{tag}_{k}"] = np.asarray(v)
        out[f"temp_after_{tag}"] = np.array([tf.temperature])
    out["volumes"] = np.array([params["volume"], small["volume"]])
    out["ku"] = np.array([params["uniaxial_anisotropy"]])
    save("G15_thermal_scalars", **out)


def G16():
    """device_helpers: analysis helpers of the SOT / VCMA device classes (power, thresholds, barriers, switching times)."""
    fac = DeviceFactory()
    out = {}
    rng = np.random.default_rng(16)
    ms_ = unit_rows(rng, 6)
    out["m"] = ms_
    sot_p = fac.get_default_parameters("sot_mram")
    sot = fac.create_device("sot_mram", sot_p)
    th = sot.get_switching_threshold()
    out["sot_threshold"] = np.array([th["critical_current_density"], th["critical_field"], th["damping_like_efficiency"], th["field_like_efficiency"]])
    js = np.array([0.0, 1e-7, 1e5, 3e6, 2e7, -5e7, 1e9])
    out["sot_J"] = js
    out["sot_power"] = np.array([sot.compute_power_consumption(j, 1e-9, ms_[0]) for j in js])
    out["sot_time"] = np.array([[sot.estimate_switching_time(j, T) for j in js] for T in (300.0, 400.0)])
    out["sot_barrier"] = np.array([sot.compute_energy_barrier(m) for m in ms_])
    vc_p = fac.get_default_parameters("vcma_mram")
    vc = fac.create_device("vcma_mram", vc_p)
    th = vc.get_switching_threshold()
    out["vcma_threshold"] = np.array([th["critical_voltage"], th["thermal_switching_voltage"], th["breakdown_voltage"], th["vcma_coefficient"]])
    vs = np.array([0.0, 1e-7, 0.05, 0.2, 0.5, -1.0, 1.9, 3.0])
    out["vcma_V"] = vs
    out["vcma_power"] = np.array([vc.compute_power_consumption(v, 1e-9) for v in vs])
    out["vcma_prob"] = np.array([[vc.compute_switching_probability(v, 1e-9, T) for v in vs] for T in (300.0, 0.0)])
    out["vcma_time"] = np.array([vc.estimate_switching_time(v, 300.0) for v in vs])
    out["vcma_barrier"] = np.array([vc.compute_energy_barrier(ms_[0], v) for v in vs])
    out["vcma_leak"] = np.array([vc.compute_leakage_current(v) for v in vs])
    out["vcma_cap"] = np.array([vc.capacitance])
    vc.update_temperature(350.0)
    th = vc.get_switching_threshold()
    out["vcma_threshold_350"] = np.array([th["critical_voltage"], th["thermal_switching_voltage"]])
    save("G16_device_helpers", **out)


def G17():
    """array_env, observation_mode='dict' (envs/array_env.py:273-287,569-578): one episode, every field of the dict."""
    from spin_torque_gym.envs.array_env import SpinTorqueArrayEnv
    rng = np.random.default_rng(1717)
    env = SpinTorqueArrayEnv(array_size=(2, 3), action_mode="individual", observation_mode="dict", max_steps=5,
                             coupling_type="dipolar", coupling_strength=0.2)
    actions = [(rng.uniform(-0.5, 5.5), rng.uniform(-2e6, 2e6), rng.uniform(1e-10, 2e-9)) for _ in range(6)]
    keys = ("current_pattern", "target_pattern", "pattern_similarity", "steps_remaining", "total_energy")
    obs, _ = env.reset(seed=17)
    rec = {k: [np.asarray(obs[k]).copy()] for k in keys}
    rec.update(reward=[], terminated=[], truncated=[])
    for a in actions:
        obs, r, te, tr, info = env.step(np.array(a, dtype=np.float32))
        for k in keys:
            rec[k].append(np.asarray(obs[k]).copy())
        rec["reward"].append(r); rec["terminated"].append(te); rec["truncated"].append(tr)
    out = {k: np.array(v) for k, v in rec.items()}
    out["actions"] = np.array(actions, dtype=np.float32)
    out["dtypes"] = np.array([str(np.asarray(obs[k]).dtype) for k in keys])
    print("    G17 dict episode:", {k: out[k].shape for k in keys}, list(out["dtypes"]))
    save("G17_array_dict", **out)


def G18():
    """find_stable_states (physics/llgs_solver.py:264-305): seeded global np.random initial states, 10 ns J = 0 relaxations."""
    solver = LLGSSolver()
    out = {}
    tags = []
    for tag, over, n_trials, seed in (("default", {}, 6, 7), ("tilted", dict(easy_axis=np.array([0.3, 0.0, 1.0])), 4, 11)):
        params = stt_params(**over)
        # the initial states the method will draw (llgs_solver.py:275-276), replayed from the same seed
        np.random.seed(seed)
        m_init = np.array([np.random.normal(0, 1, 3) for _ in range(n_trials)])
        m_init = m_init / np.linalg.norm(m_init, axis=1, keepdims=True)
        # every trial's relaxed state, through the same call the method makes (:281-288)
        finals = []
        for k in range(n_trials):
            t0 = time.time()
            r = guarded(300, solver.solve, m_init[k].copy(), (0, 10e-9), params, lambda t: 0.0, lambda t: np.zeros(3),
                        thermal_noise=False)
            finals.append(r["m"][-1].copy())
            print(f"    G18 {tag} trial {k}: {len(r['t'])} points in {time.time() - t0:.1f}s, success={r['success']}")
        # and the method itself
        np.random.seed(seed)
        states = guarded(3000, solver.find_stable_states, params, n_trials=n_trials)
        print(f"    G18 {tag}: {len(states)} stable states\n{states}")
        out[f"{tag}_seed"] = np.array(seed)
        out[f"{tag}_m_init"] = m_init
        out[f"{tag}_m_final"] = np.array(finals)
        out[f"{tag}_stable_states"] = np.asarray(states, dtype=float)
        tags.append(tag)
    out["tags"] = np.array(tags)
    save("G18_stable_states", **out)


ALL = dict(G18=G18, G17=G17, G16=G16, G15=G15, G14=G14, G1=G1, G2=G2, G3=G3, G4=G4, G5=G5, G6=G6, G7=G7, G8=G8, G9=G9, G10=G10, G11=G11, G12=G12, G13=G13)

if __name__ == "__main__":
    which = sys.argv[1:] or list(ALL)
    for w in which:
        t0 = time.time()
        print(f"{w}: {ALL[w].__doc__.strip()}")
        ALL[w]()
        print(f"  done in {time.time() - t0:.1f}s")
