"""Solver level: the `spin_torque_gym.physics` classes that sit on the step path, backed by the HIP kernels.

  LLGSSolver ............ physics/llgs_solver.py:21-305   (SciPy RK45 semantics: rtol, atol, max_step)
  SimpleLLGSSolver ...... physics/simple_solver.py:21-399 (fixed-step 'rk4' / 'euler')
  RobustLLGSSolver ...... utils/robust_solver.py:22-345   (input/output gates, fallback result)
  ThermalFluctuations ... physics/thermal_model.py:12-137 (Brown field strength, white / OU field generator)

`solve()` keeps the reference signature and result dict ('t', 'm', 'success', and for LLGSSolver 'energy').  The GPU
integrates rectangular pulses -- current_func(t) = J while t <= T, else 0, zero applied field -- which is the only
form SpinTorqueEnv ever passes (spin_torque_env.py:442-447); `solve()` recognises that form by probing the callable
and raises NotImplementedError for anything else (arbitrary Python callables cannot run in a kernel).  The batched
entry `solve_batch()` takes arrays of (m0, J, T) and is what a vectorised caller should use.

SimpleLLGSSolver here always applies RobustLLGSSolver's gates, because that is the only way the reference env runs it;
the two classes differ only in their constructor signature.  The result cache of the reference (keyed
without J, stale hits) is deliberately not reproduced.
"""
import warnings
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import torch

from .backend import EnvConfig, HipBackend
from .devices import DeviceFactory, flatten_params

MU0 = 4 * np.pi * 1e-7


class _RawParams:
    """A bare device_params dict as the reference's solvers take it: they never build a device object, they read the dict
    with `.get(key, default)` (simple_solver.py:126-131; llgs_solver.py:79-82,192-205), so a partial dict is legal there."""
    device_type = "stt_mram"

    def __init__(self, device_params: Dict[str, Any]):
        self.device_params = device_params


def _flat_for(device_params: Dict[str, Any], device_type: str = "stt_mram", validate: bool = True):
    """The C-ABI parameter record of one solve.  validate: keep the outcome of the reference's `validate_parameters(params,
    'stt_mram')` gate (RobustLLGSSolver: an invalid dict ends in the fallback result, robust_solver.py:108-110,140-150); the plain solvers
    have no such gate."""
    if device_type == "stt_mram":
        p = flatten_params(_RawParams(device_params))
    else:
        p = flatten_params(DeviceFactory().create_device(device_type, device_params))
    if not validate:
        p.params_valid = 1
    return p


def _pulse_from_callable(current_func: Optional[Callable], t0: float, t1: float) -> float:
    """Recovers J from a rectangular-pulse current function; refuses anything else."""
    if current_func is None:
        return 0.0
    j0 = float(current_func(t0))
    span = t1 - t0
    probes = [t0 + f * span for f in (0.0, 0.25, 0.5, 0.75, 1.0)]
    if any(float(current_func(t)) != j0 for t in probes):
        raise NotImplementedError("the GPU solver integrates rectangular pulses (constant J over the time span); "
                                  "use the reference CPU solver for arbitrary current_func callables")
    return j0


def _check_zero_field(field_func: Optional[Callable], t0: float, t1: float) -> None:
    if field_func is None:
        return
    for t in (t0, 0.5 * (t0 + t1), t1):
        if np.any(np.asarray(field_func(t), dtype=float) != 0.0):
            raise NotImplementedError("an applied field is not part of the SpinTorque-v0 step path "
                                      "(spin_torque_env.py:446-447 passes zeros); use the reference CPU solver")


class _GpuSolverBase:
    _solver_name = "rk4"
    _validate_params = False        # (RobustLLGSSolver turns the parameter gate on)

    def __init__(self, rtol, atol, max_step, gamma=2.21e5, device_index=0, backend=None, seed=0):
        self.rtol, self.atol, self.max_step, self.gamma = rtol, atol, max_step, gamma
        self.mu_0 = MU0
        self.device_index = device_index
        self._backend_factory = HipBackend if backend is None else backend
        self._seed = seed
        self.solve_count = 0

    def _backend(self, n, device_params, device_type, thermal_noise, temperature):
        cfg = EnvConfig(solver=self._solver_name, include_thermal_fluctuations=bool(thermal_noise),
                        temperature=float(temperature), gamma=self.gamma, max_step=self.max_step, rtol=self.rtol,
                        atol=self.atol, seed=self._seed)
        b = self._backend_factory(n, cfg, self.device_index, 0)
        b.set_params([_flat_for(device_params, device_type, self._validate_params)])
        return b

    def solve_batch(self, m_initial, current, duration, device_params: Dict[str, Any], thermal_noise: bool = False,
                    temperature: float = 300.0, device_type: str = "stt_mram", return_trajectory: int = 0,
                    env_step: int = 0) -> Dict[str, Any]:
        """N independent solves.  m_initial [N,3]; current, duration [N] (rectangular pulses over (0, duration)).
        return_trajectory = K > 0 additionally returns the first K accepted points ('t' [K,N], 'm' [K,N,3])."""
        m0 = torch.as_tensor(np.asarray(m_initial, dtype=np.float64) if not torch.is_tensor(m_initial) else m_initial)
        n = m0.shape[0]
        b = self._backend(n, device_params, device_type, thermal_noise, temperature)
        try:
            out = b.solve(m0.t().contiguous(), torch.as_tensor(current, dtype=torch.float64),
                          torch.as_tensor(duration, dtype=torch.float64), env_step=env_step,
                          traj_cap=int(return_trajectory), want_energy=self._solver_name == "rk45")
            res = {"m_final": out["m_final"].t().cpu().numpy(), "success": out["success"].cpu().numpy().astype(bool),
                   "n_points": out["n_points"].cpu().numpy()}
            if return_trajectory:
                res["t"] = out["t"].cpu().numpy()
                res["m"] = out["m"].permute(0, 2, 1).cpu().numpy()
                if out.get("energy") is not None:
                    res["energy"] = out["energy"].cpu().numpy()
                if out.get("torques") is not None:
                    res["torques"] = out["torques"].cpu().numpy()
        finally:
            b.close()
        self.solve_count += n
        return res

    def _solve_one(self, m_initial, t_span, device_params, current_func, field_func, thermal_noise, temperature,
                   traj_cap):
        t0, t1 = float(t_span[0]), float(t_span[1])
        if t0 != 0.0:
            raise NotImplementedError("time spans start at 0 on the step path (spin_torque_env.py:453)")
        J = _pulse_from_callable(current_func, t0, t1)
        _check_zero_field(field_func, t0, t1)
        r = self.solve_batch(np.asarray(m_initial, dtype=np.float64)[None, :], [J], [t1], device_params,
                             thermal_noise, temperature, return_trajectory=traj_cap)
        k = int(r["n_points"][0]) + 1
        return r, k


class SimpleLLGSSolver(_GpuSolverBase):
    """simple_solver.py:21-399 ('euler' | 'rk4'), as RobustLLGSSolver runs it."""

    def __init__(self, method: str = "euler", rtol: float = 1e-3, atol: float = 1e-6, max_step: float = 1e-12,
                 timeout: float = 2.0, **kw):
        method = method.lower()
        if method not in ("euler", "rk4"):
            warnings.warn(f"Unknown method '{method}', using 'euler'")      # simple_solver.py:54-56
            method = "euler"
        super().__init__(rtol, atol, max_step, **kw)
        self.method = method
        self._solver_name = method
        self.timeout = timeout          # kept for API compatibility; kernels need no wall-clock guard

    def solve(self, m_initial, time_span: Tuple[float, float], device_params: Dict[str, Any],
              current_func: Optional[Callable] = None, field_func: Optional[Callable] = None,
              thermal_noise: bool = False, temperature: float = 300.0) -> Dict[str, Any]:
        cap = 5002                                                           # n <= 5000 for the env's 5 ns maximum
        r, k = self._solve_one(m_initial, time_span, device_params, current_func, field_func, thermal_noise,
                               temperature, cap)
        ok = bool(r["success"][0])
        if not ok:      # robust_solver.py:278-299 fallback result: the initial state repeated
            t1 = float(time_span[1])
            npts = max(2, int(t1 / self.max_step))
            return {"t": np.linspace(0.0, t1, npts), "m": np.tile(np.asarray(m_initial, dtype=float), (npts, 1)),
                    "success": False, "message": "Fallback result", "solve_time": 0.0, "is_fallback": True}
        k = min(k, cap)
        return {"t": r["t"][:k, 0], "m": r["m"][:k, 0, :], "success": True,
                "message": "Integration completed successfully", "solve_time": 0.0, "n_steps": k - 1}

    def get_solver_info(self):
        return {"method": self.method, "solve_count": self.solve_count, "timeout_count": 0, "last_solve_time": 0.0,
                "timeout_rate": 0.0, "avg_solve_time": 0.0}


class RobustLLGSSolver(SimpleLLGSSolver):
    """utils/robust_solver.py:22-345: same kernels (the gates are always on), reference constructor signature."""
    _validate_params = True

    def __init__(self, method: str = "euler", rtol: float = 1e-3, atol: float = 1e-6, max_step: float = 1e-12,
                 timeout: float = 2.0, max_retries: int = 3, fallback_method: str = "euler",
                 enable_monitoring: bool = True, enable_validation: bool = True, **kw):
        super().__init__(method, rtol, atol, max_step, timeout, **kw)
        if not enable_validation:
            warnings.warn("enable_validation=False is not available on the GPU path; the gates stay on")
        self.max_retries, self.fallback_method = max_retries, fallback_method
        self.stats = {"total_solves": 0, "successful_solves": 0, "failed_solves": 0}

    def solve(self, m_initial, t_span, device_params, current_func=None, field_func=None, thermal_noise=False,
              temperature=300.0, **kwargs):
        res = super().solve(m_initial, t_span, device_params, current_func, field_func, thermal_noise, temperature)
        self.stats["total_solves"] += 1
        self.stats["successful_solves" if res["success"] else "failed_solves"] += 1
        return res

    def reset_statistics(self) -> None:
        """robust_solver.py:330-345 (the retry/fallback counters stay at zero here: the kernels have no retry path)."""
        self.stats = {"total_solves": 0, "successful_solves": 0, "failed_solves": 0}

    def get_statistics(self):
        st = dict(self.stats)
        tot = max(st["total_solves"], 1)
        st["success_rate"] = st["successful_solves"] / tot
        st["failure_rate"] = st["failed_solves"] / tot
        return st


class LLGSSolver(_GpuSolverBase):
    """physics/llgs_solver.py:21-305 with method='RK45' (the only method the step path configures, config.py:18-24)."""

    _solver_name = "rk45"

    def __init__(self, method: str = "RK45", rtol: float = 1e-6, atol: float = 1e-9, max_step: float = 1e-12,
                 gamma: float = 2.21e5, **kw):
        if method != "RK45":
            raise NotImplementedError(f"method '{method}': the GPU path implements SciPy's RK45 (Dormand-Prince 5(4))")
        super().__init__(rtol, atol, max_step, gamma, **kw)
        self.method = method
        self.k_b = 1.380649e-23

    def solve(self, m_initial, time_span, device_params, current_func, field_func=None, thermal_noise: bool = True,
              temperature: float = 300.0, max_points: int = 8192) -> Dict[str, Any]:
        r, k = self._solve_one(m_initial, time_span, device_params, current_func, field_func, thermal_noise,
                               temperature, max_points)
        if k > max_points:
            warnings.warn(f"trajectory truncated to max_points={max_points} of {k} accepted points")
            k = max_points
        t = r["t"][:k, 0]
        m = r["m"][:k, 0, :]
        # energy and |tau_stt| + |tau_fl| per accepted point are recorded by the kernel (llgs_solver.py:154-172)
        return {"t": t, "m": m, "energy": r["energy"][:k, 0], "torques": r["torques"][:k, 0], "success": bool(r["success"][0])}

    def find_stable_states(self, device_params: Dict[str, Any], n_trials: int = 100, threshold: float = 1e-6,
                           relax_time: float = 10e-9, seed: Optional[int] = None, initial_states=None) -> np.ndarray:
        """llgs_solver.py:264-305 as ONE batched relaxation of the n_trials random initial states (J = 0, no field, thermal
        off, 10 ns).  The reference draws each trial's state from the GLOBAL legacy generator -- `np.random.normal(0, 1, 3)`
        per trial (llgs_solver.py:275-276) -- so does this method when `seed` is None: after `np.random.seed(s)` both give
        the same initial states, hence (within the solver tolerance) the same de-duplicated list, in the same order.
        `seed` draws from a private `RandomState(seed)` instead (same stream as `np.random.seed(seed)`, global state
        untouched); `initial_states` [n,3] bypasses the draws."""
        if initial_states is not None:
            m0 = np.asarray(initial_states, dtype=np.float64).reshape(-1, 3)
        else:
            gen = np.random if seed is None else np.random.RandomState(seed)
            m0 = np.array([gen.normal(0, 1, 3) for _ in range(n_trials)], dtype=np.float64).reshape(-1, 3)
        m0 = m0 / np.linalg.norm(m0, axis=1, keepdims=True)
        n = len(m0)
        r = self.solve_batch(m0, np.zeros(n), np.full(n, relax_time), device_params, thermal_noise=False)
        states = []
        for mf, ok in zip(r["m_final"], r["success"]):          # llgs_solver.py:290-300: first-come de-duplication
            if ok and all(np.linalg.norm(mf - s) >= threshold for s in states):
                states.append(mf)
        return np.array(states) if states else np.array([[0, 0, 1]])


class ThermalFluctuations:
    """physics/thermal_model.py:12-137.  Closed-form scalars and a host-side generator seeded like the reference's
    (``np.random.default_rng(seed)``), so the same seed gives the same field samples as the reference class.  It is API
    surface only: the env never samples this object (spin_torque_env.py:103-107 builds it, :303 only sets its
    temperature); the field the step path uses is drawn inside the kernels (csrc/stg_physics.hpp: NormalStream)."""

    def __init__(self, temperature: float = 300.0, correlation_time: float = 1e-12, seed: Optional[int] = None):
        self.temperature = temperature
        self.correlation_time = correlation_time
        self.k_b = 1.380649e-23
        self.mu_0 = MU0
        self.rng = np.random.default_rng(seed)
        self._previous_noise = np.zeros(3)

    def set_temperature(self, temperature: float) -> None:
        self.temperature = temperature

    def compute_noise_strength(self, damping, saturation_magnetization, volume, gamma: float = 2.21e5) -> float:
        if self.temperature <= 0:
            return 0.0
        return np.sqrt(2 * damping * self.k_b * self.temperature / (gamma * self.mu_0 * saturation_magnetization * volume))

    def generate_thermal_field(self, damping, saturation_magnetization, volume, dt, gamma: float = 2.21e5,
                               correlated: bool = True) -> np.ndarray:
        s = self.compute_noise_strength(damping, saturation_magnetization, volume, gamma)
        if s == 0:
            return np.zeros(3)
        white = self.rng.normal(0, 1, 3)
        if correlated and self.correlation_time > 0:      # Ornstein-Uhlenbeck update, thermal_model.py:113-137
            decay = np.exp(-dt / self.correlation_time)
            self._previous_noise = decay * self._previous_noise + np.sqrt(1 - decay ** 2) * white
            return s * self._previous_noise
        return s * white

    def compute_thermal_barrier(self, anisotropy_constant: float, volume: float) -> float:
        if self.temperature <= 0:
            return float("inf")
        return anisotropy_constant * volume / (self.k_b * self.temperature)

    def compute_switching_probability(self, energy_barrier, attempt_frequency: float = 1e9,
                                      measurement_time: float = 1e-9) -> float:
        if self.temperature <= 0:
            return 0.0
        rate = attempt_frequency * np.exp(-energy_barrier / (self.k_b * self.temperature))
        return min(1 - np.exp(-rate * measurement_time), 1.0)

    # -- Neel-Brown scalars (thermal_model.py:185-336): closed forms, host only ------------------------------------------
    _YEAR = 365.25 * 24 * 3600

    def _rate(self, energy_barrier, attempt_frequency):
        return attempt_frequency * np.exp(-energy_barrier / (self.k_b * self.temperature))

    def sample_switching_time(self, energy_barrier, attempt_frequency: float = 1e9) -> float:
        """thermal_model.py:185-207: exponential waiting time at the Arrhenius rate, from this object's generator."""
        if self.temperature <= 0:
            return float("inf")
        rate = self._rate(energy_barrier, attempt_frequency)
        return float("inf") if rate <= 0 else self.rng.exponential(1.0 / rate)

    def compute_retention_time(self, energy_barrier, failure_rate: float = 1e-9, attempt_frequency: float = 1e9) -> float:
        """thermal_model.py:209-232: -ln(failure_rate) / (f0 exp(-E_b / k_B T))."""
        if self.temperature <= 0 or failure_rate <= 0:
            return float("inf")
        return -np.log(failure_rate) / (attempt_frequency * np.exp(-(energy_barrier / (self.k_b * self.temperature))))

    def analyze_thermal_stability(self, device_params: dict, time_scale: float = 10.0) -> dict:
        """thermal_model.py:234-272 (time_scale in years)."""
        volume = device_params.get("volume", 1e-24)
        k_u = device_params.get("uniaxial_anisotropy", 1e6)
        e_b = k_u * volume
        delta = self.compute_thermal_barrier(k_u, volume)
        return {"thermal_stability_factor": delta, "energy_barrier_J": e_b,
                "energy_barrier_kT": e_b / (self.k_b * self.temperature),
                "switching_probability": self.compute_switching_probability(e_b, measurement_time=time_scale * self._YEAR),
                "retention_time_years": self.compute_retention_time(e_b) / self._YEAR,
                "is_thermally_stable": delta > 40, "temperature_K": self.temperature}

    def generate_temperature_sweep(self, temp_range: Tuple[float, float], device_params: dict, n_points: int = 100) -> dict:
        """thermal_model.py:274-336: stability factor, one-year switching probability, retention (years) and Brown field
        strength over a temperature grid; the object's temperature is restored afterwards."""
        temps = np.linspace(temp_range[0], temp_range[1], n_points)
        keep = self.temperature
        volume = device_params.get("volume", 1e-24)
        k_u = device_params.get("uniaxial_anisotropy", 1e6)
        damping = device_params.get("damping", 0.01)
        ms = device_params.get("saturation_magnetization", 800e3)
        cols = {"thermal_stability_factor": [], "switching_probability": [], "retention_time": [], "noise_strength": []}
        for t in temps:
            self.set_temperature(t)
            e_b = k_u * volume
            cols["thermal_stability_factor"].append(self.compute_thermal_barrier(k_u, volume))
            cols["switching_probability"].append(self.compute_switching_probability(e_b, measurement_time=self._YEAR))
            cols["retention_time"].append(self.compute_retention_time(e_b) / self._YEAR)
            cols["noise_strength"].append(self.compute_noise_strength(damping, ms, volume))
        self.set_temperature(keep)
        out = {"temperature": temps}
        out.update({k: np.array(v) for k, v in cols.items()})
        return out
