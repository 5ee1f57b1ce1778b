"""Environment level: the Gymnasium-facing mirror of the reference's SpinTorqueEnv.

  SpinTorqueVecEnv ... N independent SpinTorque-v0 environments stepped by one kernel launch; torch tensors in
                       and out (VectorEnv-shaped API).  This is the product.
  SpinTorqueEnv ...... the reference's single-env class as a thin N=1 facade over the same kernels: same
                       constructor keywords, reset/step/render/close, info keys and error behaviour as
                       spin_torque_gym/envs/spin_torque_env.py:26-745, NumPy in and out.

Nothing here computes physics: action clamping, integration, energy, observation, reward and termination all
happen in libspintorque_hip.so.  The host side keeps what is inherently host-side in the reference too:
the seeded PCG64 generator of reset() (gymnasium.utils.seeding), the episode history list and rendering.
"""
import time
import warnings
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .backend import EnvConfig, HipBackend
from .devices import DeviceFactory, flatten_params

try:  # Gymnasium is optional (it is not part of the build image); the API shape is kept either way.
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    _EnvBase = _gym.Env
except Exception:  # pragma: no cover - exercised in the build image
    _gym = None
    _spaces = None
    _EnvBase = object


class _Box:
    """Minimal stand-in for gymnasium.spaces.Box when Gymnasium is absent (attributes only)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        self.shape = tuple(shape) if shape is not None else low.shape
        self.low = np.broadcast_to(low, self.shape).astype(dtype)
        self.high = np.broadcast_to(high, self.shape).astype(dtype)
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains           # `obs in env.observation_space`, as with gymnasium.spaces.Box


def _box(low, high, shape=None, dtype=np.float32):
    if _spaces is not None:
        return _spaces.Box(low=np.asarray(low, dtype=dtype) if shape is None else low,
                           high=np.asarray(high, dtype=dtype) if shape is None else high, shape=shape, dtype=dtype)
    return _Box(low, high, shape, dtype)


def _np_random(seed=None):
    """gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed)))."""
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy


class _TimerTable:
    """Host-side timer table behind get_performance_stats() (the reference's PerformanceProfiler.get_stats shape,
    utils/performance.py:391-474: `<op>_avg_time/_total_time/_count/_min_time/_max_time` plus named counters).  Times are
    host wall-clock around the call: for the vector env that is the enqueue cost of an asynchronous launch, for the N = 1
    facade (which reads the state back every step) the whole step."""

    def __init__(self):
        self._t: Dict[str, List[float]] = {}
        self.counters: Dict[str, int] = {}

    def add(self, name: str, seconds: float) -> None:
        r = self._t.setdefault(name, [0, 0.0, float("inf"), 0.0])
        r[0] += 1
        r[1] += seconds
        r[2] = min(r[2], seconds)
        r[3] = max(r[3], seconds)

    def get_stats(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        for name, (cnt, tot, lo, hi) in self._t.items():
            out.update({f"{name}_avg_time": tot / cnt, f"{name}_total_time": tot, f"{name}_count": cnt,
                        f"{name}_min_time": lo, f"{name}_max_time": hi})
        out.update(self.counters)
        return out

    def reset(self) -> None:
        self._t.clear()
        self.counters.clear()


def _default_env_side_params(device_type: str) -> Dict[str, Any]:
    """SpinTorqueEnv._get_default_device_params (spin_torque_env.py:156-182): the factory's STT defaults, or a
    generic dict for other types (which lacks 'easy_axis', so the factory then rejects it, as in the reference)."""
    if device_type == "stt_mram":
        return DeviceFactory().get_default_parameters("stt_mram")
    return {"volume": 1e-24, "saturation_magnetization": 800e3, "damping": 0.01, "uniaxial_anisotropy": 1e6,
            "polarization": 0.7}


class SpinTorqueVecEnv:
    """N parallel SpinTorque-v0 environments on one MI355X.

    Device classes: pass one ``device_type``/``device_params`` for a homogeneous batch, or lists of both plus
    ``class_index`` (uint8 per env) for mixed batches (up to 64 classes; the per-class constants live in LDS).
    Layout: with ``out_layout='records'`` (default) the kernel writes one 56-byte record per env -- obs[12] f32, reward f32,
    terminated, truncated, status -- and ``obs`` / ``reward`` / the flags are strided *views* of that array (``obs`` is [N,12],
    row stride 14 floats); ``out_layout='soa'`` keeps four separate arrays with a component-major [12,N] obs buffer, of which
    ``obs`` is the transposed view.  No copies either way.  Actions are accepted as [N,2] (Gym convention) or, with
    ``actions_soa=True``, as the kernel's [2,N].
    With ``autoreset`` a step also returns ``info['final_obs']`` -- the terminal observation of the envs whose episode ended on
    this step (their ``obs`` row already holds the new episode's first observation).  ``diagnostics=True`` additionally fills
    ``info['reward_f64']`` (the reward before rounding to fp32) and ``info['energy']`` (the reference's
    info['energy_consumed'], spin_torque_env.py:474-480); by default a step writes the RL-facing outputs only
    (``info['status']`` is then the records' status byte).
    """

    def __init__(self, num_envs: int, device_type: Union[str, Sequence[str]] = "stt_mram",
                 device_params: Union[None, Dict[str, Any], Sequence[Dict[str, Any]]] = None,
                 class_index=None, target_states: Optional[List[np.ndarray]] = None, max_steps: int = 100,
                 max_current: float = 2e6, max_duration: float = 5e-9, temperature: float = 300.0,
                 include_thermal_fluctuations: bool = True, success_threshold: float = 0.9,
                 energy_penalty_weight: float = 0.1, solver: str = "rk4", seed: Optional[int] = None,
                 autoreset: bool = False, skip_done: bool = False, device_index: int = 0, env_id0: int = 0,
                 max_attempts: int = 200_000, lane_sort: Optional[bool] = None, wave_spec: Optional[bool] = None, torque_model: str = "reference",
                 noise_model: str = "white", correlation_time: float = 1e-12,
                 per_env_params: Optional[Dict[str, Any]] = None, out_layout: str = "records",
                 diagnostics: bool = False, lane_refill: Optional[int] = None, backend=None):
        self.num_envs = int(num_envs)
        factory = DeviceFactory()
        types = [device_type] if isinstance(device_type, str) else list(device_type)
        if device_params is None:
            plist = [_default_env_side_params(t) for t in types]
        elif isinstance(device_params, dict):
            plist = [device_params]
        else:
            plist = list(device_params)
        if len(plist) != len(types):
            raise ValueError("device_type and device_params must have the same length")
        self.devices = [factory.create_device(t, p) for t, p in zip(types, plist)]
        self.device_types = types
        if target_states is None:
            targets = [np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, -1.0])]
        else:
            targets = [self.devices[0].validate_magnetization(np.asarray(t, dtype=float)) for t in target_states]
        self.target_states = targets
        self._rng, self._seed = _np_random(seed)
        self.cfg = EnvConfig(solver=solver, include_thermal_fluctuations=include_thermal_fluctuations,
                             temperature=temperature, max_steps=max_steps, max_current=max_current,
                             max_duration=max_duration, success_threshold=success_threshold,
                             energy_penalty_weight=energy_penalty_weight, target_states=[list(t) for t in targets],
                             seed=int(self._rng.integers(0, 2**63 - 1)) if seed is None else int(seed),
                             max_attempts=max_attempts, skip_done=skip_done, lane_sort=lane_sort, wave_spec=wave_spec,
                             torque_model=torque_model, noise_model=noise_model, correlation_time=correlation_time,
                             out_layout=out_layout, diagnostics=bool(diagnostics), lane_refill=lane_refill)
        self.autoreset = bool(autoreset)
        self.diagnostics = bool(diagnostics)
        # `backend` is a test seam: a class/callable with HipBackend's constructor signature (tests inject the CPU
        # oracle for the gloo runs and as the comparator); the product default is the HIP library, nothing else.
        self._backend_factory = HipBackend if backend is None else backend
        self._device_index, self.env_id0 = int(device_index), int(env_id0)
        self._class_index, self._per_env_params = class_index, per_env_params
        if per_env_params and len(self.devices) > 1 and class_index is None:
            raise ValueError("per_env_params with several base device classes needs class_index (the base of each env)")
        self.profiler = _TimerTable()
        self.backend = self._make_backend()
        self.single_action_space = _box([-max_current, 0.0], [max_current, max_duration], dtype=np.float32)
        self.single_observation_space = _box(-np.inf, np.inf, shape=(12,), dtype=np.float32)
        self._needs_reset = True

    # batched spaces of gymnasium.vector.VectorEnv, built on first use (N x 2 / N x 12 bounds)
    @property
    def action_space(self):
        if getattr(self, "_action_space", None) is None:
            lo, hi = self.single_action_space.low, self.single_action_space.high
            self._action_space = _box(np.tile(lo, (self.num_envs, 1)), np.tile(hi, (self.num_envs, 1)), dtype=np.float32)
        return self._action_space

    @property
    def observation_space(self):
        if getattr(self, "_observation_space", None) is None:
            self._observation_space = _box(-np.inf, np.inf, shape=(self.num_envs, 12), dtype=np.float32)
        return self._observation_space

    def _make_backend(self):
        """One context for (num_envs, cfg, env_id0) with this env's device parameters installed."""
        b = self._backend_factory(self.num_envs, self.cfg, self._device_index, self.env_id0)
        if self._per_env_params:
            # device-to-device variation: every env gets its own record, starting from its base device class (one class, or
            # class_index into several -- a mixed STT/SOT/VCMA batch); keys are the reference's device_params keys, values
            # arrays of length num_envs ([num_envs, 3] for vectors)
            from .devices import per_env_param_block_multi
            cls = None if self._class_index is None else torch.as_tensor(self._class_index).cpu().numpy()
            b.set_params_per_env(*per_env_param_block_multi([flatten_params(d) for d in self.devices], cls, self.num_envs,
                                                            self._per_env_params))
        else:
            b.set_params([flatten_params(d) for d in self.devices], self._class_index)
        return b

    # -- Gymnasium VectorEnv-shaped API ---------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        """options: 'initial_state' / 'target_state' as [N,3] (or [3], broadcast) arrays, 'mask' (bool [N]) to
        reset a subset.  Without them states are drawn on the device from Philox(seed, env_id, ...)."""
        options = options or {}
        if seed is not None:
            self._rng, self._seed = _np_random(seed)
        dev_seed = int(self._rng.integers(0, 2**63 - 1))
        init = self._soa3(options.get("initial_state"))
        tgt = self._soa3(options.get("target_state"))
        mask = options.get("mask")
        if mask is not None:
            mask = torch.as_tensor(mask).to(torch.uint8)
        t0 = time.perf_counter()
        obs = self.backend.reset(mask, init, tgt, dev_seed)
        self.profiler.add("reset", time.perf_counter() - t0)
        self._needs_reset = False
        return obs.t(), {}

    def step(self, actions, actions_soa: bool = False, out=None):
        """out (out_layout='records' only): a uint8 [N,56] record array that receives this step's outputs instead of the
        env's own (double buffering; the multi-GPU env passes its slice of the global record array)."""
        if self._needs_reset:
            raise RuntimeError("Environment must be reset before calling step")
        a = torch.as_tensor(actions)
        if not actions_soa:
            a = a.t()
        t0 = time.perf_counter()
        if out is None:
            obs, rew, rew64, term, trunc, status = self.backend.step(a, autoreset=self.autoreset)
        else:
            obs, rew, rew64, term, trunc, status = self.backend.step(a, autoreset=self.autoreset, out=out)
        self.profiler.add("step", time.perf_counter() - t0)
        # diagnostics=False (default): the launch writes the RL-facing outputs only; `status` is then the status byte of the
        # records (a view; absent in the SoA layout).  diagnostics=True adds the fp64 reward and the step's Joule energy (the
        # reference's info['energy_consumed'] etc. at N = 1).
        info = {} if status is None else {"status": status}
        if self.autoreset:
            # same-step auto-reset: rows of `obs` whose episode just ended already hold the new episode's first
            # observation; their terminal observation is in info["final_obs"] (valid where terminated | truncated) --
            # RL-facing data (bootstrapping at truncation), so it is there whatever `diagnostics` says
            info["final_obs"] = self.backend.final_obs.t()
        if self.diagnostics:
            info.update(reward_f64=rew64, energy=self.backend.energy)
        return obs.t(), rew, term.bool(), trunc.bool(), info

    def step_many(self, actions, out_every: bool = True, actions_soa: bool = False):
        """K env steps in one launch.  actions [K,N,2] (or [K,2,N] with actions_soa)."""
        a = torch.as_tensor(actions)
        if not actions_soa:
            a = a.transpose(1, 2)
        t0 = time.perf_counter()
        obs, rew, rew64, term, trunc, status = self.backend.step_many(a, out_every=out_every, autoreset=self.autoreset)
        self.profiler.add("step_many", time.perf_counter() - t0)
        info = {} if status is None else {"status": status}
        if self.autoreset:
            info["final_obs"] = self.backend.final_obs_many.transpose(1, 2)     # [K or 1, N, 12]
        if self.diagnostics:
            info.update(reward_f64=rew64, energy=self.backend.energy_many)
        return obs.transpose(1, 2), rew, term.bool(), trunc.bool(), info

    def _soa3(self, v):
        if v is None:
            return None
        t = torch.as_tensor(np.asarray(v, dtype=np.float64) if not torch.is_tensor(v) else v).to(torch.float64)
        if t.dim() == 1:
            t = t.unsqueeze(0).expand(self.num_envs, 3)
        if tuple(t.shape) != (self.num_envs, 3):
            raise ValueError(f"expected [N,3] or [3], got {tuple(t.shape)}")
        # device.validate_magnetization (base_device.py:108-116) raises for a zero vector; the kernel then normalises
        if not bool(torch.all(torch.linalg.norm(t, dim=1) >= 1e-12)) or not bool(torch.isfinite(t).all()):
            raise ValueError("Magnetization vector cannot be zero")
        return t.t().contiguous()

    # -- checkpoint / resume ---------------------------------------------------------------------------
    _HOST_KEYS = ("host_rng", "cfg_seed", "env_id0")

    def state_dict(self):
        """Device state + everything the random streams hang on: the host PCG64 state (reset seeds), `cfg.seed` (the
        Philox key of the thermal field and of device-side auto-resets) and `env_id0` (its counter offset)."""
        st = {k: v.cpu() for k, v in self.backend.get_state().items()}
        st["host_rng"] = self._rng.bit_generator.state
        st["cfg_seed"] = int(self.cfg.seed)
        st["env_id0"] = int(self.env_id0)
        return st

    def load_state_dict(self, st):
        """Resumes bit-for-bit: an env built with another stream key (e.g. seed=None in a new process) or env_id0 gets
        its context rebuilt with the checkpoint's before the state is restored."""
        seed, id0 = int(st.get("cfg_seed", self.cfg.seed)), int(st.get("env_id0", self.env_id0))
        if seed != int(self.cfg.seed) or id0 != self.env_id0:
            self.backend.close()
            self.cfg.seed, self.env_id0 = seed, id0
            self.backend = self._make_backend()
        self.backend.set_state({k: v for k, v in st.items() if k not in self._HOST_KEYS})
        if "host_rng" in st:
            self._rng.bit_generator.state = st["host_rng"]
        self._needs_reset = False

    def get_performance_stats(self) -> Dict[str, Any]:
        """Host timer table + on-device counters (the reference's get_performance_stats shape, spin_torque_env.py:711-718;
        its 'optimizer' entry is the result/observation cache, which is deliberately not reproduced -- SURVEY H1/H2)."""
        c = self.backend.counters()
        prof = self.profiler.get_stats()
        prof.update(env_steps=c["env_steps"], solver_work_units=c["work_units"], noop_steps=c["noop_steps"])
        return {"profiler": prof, "optimizer": {"cache": "not reproduced (SURVEY H1/H2)", "cache_hits": 0, "cache_misses": 0},
                "health": self.get_health_report()}

    def get_state(self):
        return self.backend.get_state()

    def get_health_report(self):
        from .harness import health_report
        return health_report(self)

    def close(self):
        self.backend.close()


class SpinTorqueEnv(_EnvBase):
    """Drop-in for spin_torque_gym.envs.SpinTorqueEnv (spin_torque_env.py:26-745) running on the GPU path.

    Same keyword arguments; two additions select what the reference cannot express: ``solver`` ('rk4' is the
    SimpleLLGSSolver/RobustLLGSSolver pair the reference env really uses, 'rk45' the LLGSSolver the north star
    names) and ``device_index``.  Known reference behaviours are reproduced, not repaired:
    solver failure leaves m unchanged but still charges energy (H3), the energy "penalty" is a bonus (H7),
    action_mode='discrete' and observation_mode='dict' end in the catch-all error return (H8).  Not reproduced:
    the result/observation caches (H1/H2), which return stale data.
    """

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 30}

    def __init__(self, device_type: str = "stt_mram", device_params: Optional[Dict[str, Any]] = None,
                 target_states: Optional[List[np.ndarray]] = None, max_steps: int = 100, max_current: float = 2e6,
                 max_duration: float = 5e-9, temperature: float = 300.0, include_thermal_fluctuations: bool = True,
                 reward_components: Optional[Dict[str, Dict]] = None, action_mode: str = "continuous",
                 observation_mode: str = "vector", success_threshold: float = 0.9, energy_penalty_weight: float = 0.1,
                 render_mode: Optional[str] = None, seed: Optional[int] = None, solver: str = "rk4",
                 device_index: int = 0, backend=None):
        if reward_components is not None:
            raise NotImplementedError("custom reward callables are arbitrary Python and stay on the reference "
                                      "(SURVEY.md section 2); the GPU path implements the default 4-component reward")
        if action_mode not in ("continuous", "discrete"):
            raise ValueError(f"Unknown action mode: {action_mode}")
        if observation_mode not in ("vector", "dict"):
            raise ValueError(f"Unknown observation mode: {observation_mode}")
        self.device_type = device_type
        self.max_steps, self.max_current, self.max_duration = max_steps, max_current, max_duration
        self.temperature, self.include_thermal = temperature, include_thermal_fluctuations
        self.action_mode, self.observation_mode = action_mode, observation_mode
        self.success_threshold, self.energy_penalty_weight = success_threshold, energy_penalty_weight
        self.render_mode = render_mode
        self._np_random = None
        self.seed(seed)
        self._vec = SpinTorqueVecEnv(1, device_type, device_params, None, target_states, max_steps, max_current,
                                     max_duration, temperature, include_thermal_fluctuations, success_threshold,
                                     energy_penalty_weight, solver, seed, False, False,
                                     device_index, 0, diagnostics=True, backend=backend)
        self.device = self._vec.devices[0]
        self.target_states = self._vec.target_states
        self.solver_name = solver
        if action_mode == "continuous":
            self.action_space = _box([-max_current, 0.0], [max_current, max_duration], dtype=np.float32)
        else:
            self.current_levels = np.linspace(-max_current, max_current, 5)
            self.duration_levels = np.array([0.1e-9, 0.5e-9, 1.0e-9, 2.0e-9])
            self.action_space = _spaces.Discrete(20) if _spaces is not None else None
        self.observation_space = _box(-np.inf, np.inf, shape=(12,), dtype=np.float32)
        self.current_magnetization = None
        self.target_magnetization = None
        self.step_count = 0
        self.total_energy = 0.0
        self.episode_history: List[Dict[str, Any]] = []
        self.last_action = np.zeros(2)
        self._solve_count = 0
        self._solve_time = [0.0, 0.0]         # last, total (wall time of the synchronous N = 1 step)
        self.profiler = _TimerTable()
        self.renderer = None                  # the persistent figure of render('human') (spin_torque_env.py:152-154, 570-587)
        if render_mode == "human":
            from .render import HumanFigure
            self.renderer = HumanFigure("macrospin")

    # -- seeding (spin_torque_env.py:694-697) -------------------------------------------------------------
    def seed(self, seed: Optional[int] = None):
        self._np_random, s = _np_random(seed)
        return [s]

    # -- reset (spin_torque_env.py:250-308) -----------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        t_reset = time.perf_counter()
        if seed is not None:
            self._np_random, _ = _np_random(seed)
        options = options or {}
        self.step_count, self.total_energy = 0, 0.0
        self.episode_history = []
        self.last_action = np.zeros(2)
        if "initial_state" in options:
            m0 = self.device.validate_magnetization(options["initial_state"])
        else:
            m0 = self.device.validate_magnetization(self._np_random.normal(0, 1, 3))
        if "target_state" in options:
            tgt = self.device.validate_magnetization(options["target_state"])
        else:
            tgt = self.target_states[self._np_random.integers(len(self.target_states))].copy()
        obs, _ = self._vec.reset(options={"initial_state": np.asarray(m0, dtype=np.float64)[None, :],
                                          "target_state": np.asarray(tgt, dtype=np.float64)[None, :]})
        self._pull_state()
        self.profiler.add("env_reset", time.perf_counter() - t_reset)
        return obs[0].cpu().numpy().copy(), self._get_info()

    def _pull_state(self):
        st = self._vec.get_state()
        self.current_magnetization = st["m"][:, 0].cpu().numpy().copy()
        self.target_magnetization = st["target"][:, 0].cpu().numpy().copy()
        self.total_energy = float(st["total_energy"][0])
        self.step_count = int(st["step_count"][0])

    # -- step (spin_torque_env.py:310-407) ------------------------------------------------------------------
    def step(self, action):
        if self.current_magnetization is None:
            raise RuntimeError("Environment must be reset before calling step")
        try:
            if self.action_mode != "continuous" or self.observation_mode != "vector":
                # H8: the reference's safety wrapper rejects these modes' inputs and the catch-all takes over
                raise TypeError("only length-1 arrays can be converted to Python scalars")
            a = action if isinstance(action, np.ndarray) else np.array(action, dtype=np.float32)
            if a.shape != (2,):                     # monitoring.py:300-302
                a = np.array([0.0, 1e-12], dtype=np.float32)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float32)
            prev_alignment = float(np.dot(self.current_magnetization, self.target_magnetization))
            t_step = time.perf_counter()
            obs, rew, term, trunc, info_t = self._vec.step(torch.from_numpy(np.ascontiguousarray(a)).unsqueeze(0))
            obs_np = obs[0].cpu().numpy().copy()
            reward = float(info_t["reward_f64"][0])
            status = int(info_t["status"][0])
            energy = float(info_t["energy"][0])
            self._pull_state()
            self._solve_count += 1
            dt_step = time.perf_counter() - t_step
            self._solve_time = [dt_step, self._solve_time[1] + dt_step]
            self.profiler.add("env_step", dt_step)
            # the kernel's parsed action comes back through the observation; recompute it in fp64 for `info`
            J, T = _parse_action_host(a, self.max_current, self.max_duration)
            self.last_action = np.array([J, T])
            alignment = float(np.dot(self.current_magnetization, self.target_magnetization))
            terminated, truncated = bool(term[0]), bool(trunc[0])
            self.episode_history.append({"step": self.step_count, "action": [J, T],
                                         "magnetization": self.current_magnetization.copy(), "reward": reward,
                                         "energy": energy, "alignment": alignment})
            info = self._get_info()
            info.update({"final_magnetization": self.current_magnetization.copy(), "energy_consumed": energy,
                         "pulse_duration": T, "current_density": J,
                         "simulation_success": status != _lib.STATUS_NOOP})
            info.update({"is_success": terminated, "step_energy": energy,
                         "alignment_improvement": alignment - prev_alignment, "current_alignment": alignment})
            success_c = 10.0 if terminated else 0.0
            info.update({"reward_components": {"success": success_c, "energy": -energy / 1e-12,
                                               "progress": alignment - prev_alignment, "stability": 0.0},
                         "total_reward": reward, "solver_status": status})
            return obs_np, reward, terminated, truncated, info
        except Exception as e:                      # spin_torque_env.py:397-407
            obs = self._vec.backend.obs[:, 0].cpu().numpy().copy()
            return obs, -1.0, False, True, {"error": str(e), "step_count": self.step_count}

    def _get_info(self):
        align = float(np.dot(self.current_magnetization, self.target_magnetization)) if self.current_magnetization is not None else 0.0
        return {"step_count": self.step_count, "total_energy": self.total_energy, "current_alignment": align,
                "is_success": align >= self.success_threshold, "target_reached": align >= self.success_threshold,
                "magnetization_magnitude": float(np.linalg.norm(self.current_magnetization)) if self.current_magnetization is not None else 0.0,
                "device_type": self.device_type, "episode_history": self.episode_history.copy()}

    # -- rendering (spin_torque_env.py:556-684): host-side matplotlib, optional (spin_torque_gym_amd/render.py) -------
    def render(self, mode: Optional[str] = None):
        """'human': the reference's four-panel figure (3-D magnetisation arrows in the unit sphere; energy, alignment and current
        histories), one persistent figure redrawn in place, returns None; 'rgb_array': the x-y projection as uint8 [H, W, 3]."""
        mode = self.render_mode if mode is None else mode
        if mode is None:
            return None
        if mode not in ("human", "rgb_array"):
            raise ValueError(f"Unsupported render mode: {mode}")
        from . import render as _render
        if self.current_magnetization is None:
            raise RuntimeError("Environment must be reset before calling render")
        if mode == "rgb_array":
            return _render.macrospin_rgb(self)
        if self.renderer is None:
            self.renderer = _render.HumanFigure("macrospin")
        self.renderer.draw(self)
        return None

    def close(self):
        if self.renderer is not None:
            self.renderer.close()
            self.renderer = None
        self._vec.close()

    # -- introspection (spin_torque_env.py:699-745) ------------------------------------------------------------
    def get_device_info(self):
        return self.device.get_device_info()

    def get_solver_info(self):
        return {"method": self.solver_name, "solve_count": self._solve_count, "timeout_count": 0,
                "last_solve_time": self._solve_time[0], "timeout_rate": 0.0,
                "avg_solve_time": self._solve_time[1] / max(self._solve_count, 1), "backend": "hip/gfx950"}

    def get_health_report(self):
        from .harness import health_report
        return health_report(self)

    def get_performance_stats(self):
        """spin_torque_env.py:711-718: {'profiler', 'optimizer', 'solver', 'health'}.  The profiler table carries this
        facade's synchronous step()/reset() wall times (`env_step_*`, `env_reset_*`), the vector env's launch times and
        the device counters."""
        st = self._vec.get_performance_stats()
        prof = dict(st["profiler"])
        prof.update(self.profiler.get_stats())
        return {"profiler": prof, "optimizer": st["optimizer"], "solver": self.get_solver_info(), "health": st["health"]}

    def analyze_episode(self):
        if not self.episode_history:
            return {}
        h = self.episode_history
        total_energy = sum(x["energy"] for x in h)
        final_alignment = h[-1]["alignment"]
        switching_step = next((i + 1 for i, x in enumerate(h) if x["alignment"] >= self.success_threshold), None)
        return {"episode_length": len(h), "total_energy": total_energy, "final_alignment": final_alignment,
                "success": final_alignment >= self.success_threshold, "switching_step": switching_step,
                "average_reward": float(np.mean([x["reward"] for x in h])),
                "energy_efficiency": final_alignment / total_energy if total_energy > 0 else 0, "history": h.copy()}


def _parse_action_host(a: np.ndarray, max_current: float, max_duration: float):
    """Host restatement of the action clamp for the `info` dict only (monitoring.py:304-313,
    spin_torque_env.py:417-431); the values that drive the physics are computed in the kernel."""
    a = a.copy()
    if not np.isnan(a[0]):
        a[0] = np.clip(a[0], -1e8, 1e8)
    if not np.isnan(a[1]):
        a[1] = np.clip(a[1], 1e-12, 1e-6)
    if np.any(np.isnan(a)) or np.any(np.isinf(a)):
        a = np.array([0.0, 1e-12], dtype=a.dtype)
    J = float(np.clip(float(a[0]), -max_current, max_current))
    T = float(np.clip(float(a[1]), 1e-12, max_duration))
    return J, T


# What the reference has registered once `spin_torque_gym` and `spin_torque_gym.envs` are imported: both modules register the
# same ids, the second registration (envs/__init__.py:14-26) replaces the first (__init__.py:14-24).
REGISTRATIONS = {
    "SpinTorque-v0": dict(entry_point="spin_torque_gym_amd.envs:SpinTorqueEnv", max_episode_steps=100,
                          kwargs={"device_type": "stt_mram"}),
    "SpinTorqueArray-v0": dict(entry_point="spin_torque_gym_amd.array_env:SpinTorqueArrayEnv", max_episode_steps=200,
                               kwargs={"array_size": (4, 4), "device_type": "stt_mram"}),
}


def register_envs():
    """Registers 'SpinTorque-v0' and 'SpinTorqueArray-v0' with Gymnasium when it is installed, with the reference's final
    `max_episode_steps` and kwargs (reference: spin_torque_gym/__init__.py:14-24, envs/__init__.py:14-26), so that
    ``gym.make('SpinTorque-v0', **kwargs)`` builds the GPU-backed classes.  Called on ``import spin_torque_gym_amd``.
    ('SkyrmionRacetrack-v0' is a different physics and out of scope)"""
    if _gym is None:
        return False
    from gymnasium.envs.registration import register, registry
    for env_id, spec in REGISTRATIONS.items():
        if env_id not in registry:
            register(id=env_id, entry_point=spec["entry_point"], max_episode_steps=spec["max_episode_steps"],
                     kwargs=dict(spec["kwargs"]))
    return True
