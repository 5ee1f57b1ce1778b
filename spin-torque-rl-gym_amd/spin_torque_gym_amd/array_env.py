"""SpinTorqueArray-v0 on the GPU: host mirror of spin_torque_gym/envs/array_env.py.

  SpinTorqueArrayVecEnv ... N independent R x C device arrays per kernel launch (torch tensors in and out)
  SpinTorqueArrayEnv ...... the reference's class as an N = 1 facade: same constructor keywords, reset/step, info keys

As in the reference, a step addresses one device / one row / one column / all devices ('action_mode'), integrates the
addressed devices one after the other with ten normalised Euler sub-steps of a single derivative
(array_env.py:496-521), each seeing the updated states of its predecessors through the coupling sum, and rewards the
similarity with a target pattern.  Reference quirks are reproduced, not repaired: in 'global' mode the two-element
action's second entry (the duration) is read as the current density (array_env.py:413-414), and the energy uses the
resistance of the *updated* state.  Supported: default reward, observation modes 'array' and 'vector'.
"""
import ctypes as C
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .devices import DeviceFactory, flatten_params
from .envs import _EnvBase, _box, _np_random, _spaces

ACTION_MODES = {"individual": 0, "row": 1, "column": 2, "global": 3}
OBS_MODES = {"array": 0, "vector": 1, "dict": 1}      # 'dict' regroups the vector observation on the host side


def compute_coupling_matrix(n_rows: int, n_cols: int, coupling_type: str = "dipolar", coupling_strength: float = 0.1):
    """array_env.py:301-334 (host, once per env: an n x n table)."""
    if coupling_type not in ("dipolar", "exchange", "stray_field"):
        return np.zeros((n_rows * n_cols, n_rows * n_cols))     # the reference leaves unknown types at zero
    n = n_rows * n_cols
    idx = np.arange(n)
    r, c = np.divmod(idx, n_cols)
    dist = np.sqrt((r[:, None] - r[None, :]) ** 2 + (c[:, None] - c[None, :]) ** 2)
    out = np.zeros((n, n))
    off = dist > 0
    if coupling_type == "dipolar":
        out[off] = coupling_strength / dist[off] ** 3
    elif coupling_type == "exchange":
        out[dist == 1] = coupling_strength
    else:
        out[off] = coupling_strength / dist[off] ** 2
    return out


def checkerboard_pattern(n_rows: int, n_cols: int) -> np.ndarray:
    """array_env.py:161-170."""
    p = np.zeros((n_rows, n_cols, 3))
    ii, jj = np.meshgrid(np.arange(n_rows), np.arange(n_cols), indexing="ij")
    p[..., 2] = np.where((ii + jj) % 2 == 0, 1.0, -1.0)
    return p


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipArrayBackend:
    """One stg_array_ctx.  Per-array tensors are component-major: pattern [n_dev*3, N], obs [obs_dim, N]."""

    def __init__(self, n_arrays: int, cfg: "_lib.StgArrayConfig", dev_params: "_lib.StgDeviceParams", coupling,
                 device_index: int = 0, env_id0: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipArrayBackend needs a visible MI355X; there is no CPU fallback")
        self.n, self.cfg = int(n_arrays), cfg
        self.n_dev = cfg.rows * cfg.cols
        self.obs_dim = self.n_dev * 6 + (4 if cfg.obs_mode == 1 else 0)
        self.device = torch.device("cuda", device_index)
        self._ctx = C.c_void_p()
        cm = None
        if cfg.include_coupling:
            cm = np.ascontiguousarray(coupling, dtype=np.float64)
        _lib.check(self.lib.stg_array_create(C.byref(self._ctx), device_index, self.n, int(env_id0), C.byref(cfg),
                                             C.byref(dev_params),
                                             cm.ctypes.data_as(C.POINTER(C.c_double)) if cm is not None else None))
        n, dev = self.n, self.device
        self.obs = torch.empty((self.obs_dim, n), dtype=torch.float32, device=dev)
        self.reward = torch.empty(n, dtype=torch.float32, device=dev)
        self.reward64 = torch.empty(n, dtype=torch.float64, device=dev)
        self.energy = torch.empty(n, dtype=torch.float64, device=dev)
        self.terminated = torch.empty(n, dtype=torch.uint8, device=dev)
        self.truncated = torch.empty(n, dtype=torch.uint8, device=dev)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype, shape):
        if t is None:
            return None
        t = torch.as_tensor(t).to(device=self.device, dtype=dtype).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def close(self):
        if self._ctx:
            torch.cuda.synchronize(self.device)
            self.lib.stg_array_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None, init_pattern=None, target=None, seed=0):
        mask = self._dev(mask, torch.uint8, (self.n,))
        init_pattern = self._dev(init_pattern, torch.float64, (self.n_dev * 3, self.n))
        target = self._dev(target, torch.float64, (self.n_dev * 3, self.n))
        _lib.check(self.lib.stg_array_reset(self._ctx, _ptr(mask), _ptr(init_pattern), _ptr(target),
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(self.obs), self._stream()))
        self._keep = (mask, init_pattern, target)
        return self.obs

    def step(self, actions):
        a_dim = 2 if self.cfg.action_mode == 3 else 3
        a = self._dev(actions, torch.float32, (a_dim, self.n))
        _lib.check(self.lib.stg_array_step(self._ctx, _ptr(a), _ptr(self.obs), _ptr(self.reward), _ptr(self.reward64),
                                           _ptr(self.energy), _ptr(self.terminated), _ptr(self.truncated), self._stream()))
        self._keep = (a,)
        return self.obs, self.reward, self.reward64, self.terminated, self.truncated

    def get_state(self):
        n, dev = self.n, self.device
        st = dict(pattern=torch.empty((self.n_dev * 3, n), dtype=torch.float64, device=dev),
                  target=torch.empty((self.n_dev * 3, n), dtype=torch.float64, device=dev),
                  total_energy=torch.empty(n, dtype=torch.float64, device=dev),
                  step_count=torch.empty(n, dtype=torch.int32, device=dev))
        _lib.check(self.lib.stg_array_get_state(self._ctx, _ptr(st["pattern"]), _ptr(st["target"]), _ptr(st["total_energy"]),
                                                _ptr(st["step_count"]), self._stream()))
        return st


class SpinTorqueArrayVecEnv:
    """N parallel SpinTorqueArray-v0 environments.  Observations come back as [N, obs_dim] views ('array' mode:
    reshape to [N, rows, cols, 6]; 'dict' mode: the reference's five fields (array_env.py:569-578) as tensors with a
    leading N); actions are [N, 3] ([index, J, T]) or [N, 2] in 'global' mode."""

    def __init__(self, num_envs: int, array_size: Tuple[int, int] = (4, 4), device_type: str = "stt_mram",
                 device_params: Optional[Dict[str, Any]] = None, target_pattern: Optional[np.ndarray] = None,
                 max_steps: int = 200, max_current: float = 2e6, max_duration: float = 5e-9, temperature: float = 300.0,
                 include_coupling: bool = True, coupling_strength: float = 0.1, coupling_type: str = "dipolar",
                 action_mode: str = "individual", observation_mode: str = "array", success_threshold: float = 0.9,
                 energy_penalty_weight: float = 0.1, seed: Optional[int] = None, device_index: int = 0, env_id0: int = 0,
                 backend=None):
        if action_mode not in ACTION_MODES:
            raise ValueError(f"Unknown action mode: {action_mode}")
        if observation_mode not in OBS_MODES:
            raise ValueError(f"Unknown observation mode: {observation_mode}")
        self.num_envs = int(num_envs)
        self.array_size = tuple(array_size)
        self.n_rows, self.n_cols = self.array_size
        self.n_devices = self.n_rows * self.n_cols
        if device_params is None:
            device_params = DeviceFactory().get_default_parameters("stt_mram")     # array_env.py:152-168
        self.device = DeviceFactory().create_device(device_type, device_params)
        self.device_type, self.action_mode, self.observation_mode = device_type, action_mode, observation_mode
        self.coupling_matrix = compute_coupling_matrix(self.n_rows, self.n_cols, coupling_type, coupling_strength) \
            if include_coupling else None
        if target_pattern is not None and tuple(np.shape(target_pattern)) != (self.n_rows, self.n_cols, 3):
            raise ValueError(f"Target pattern shape must be {(self.n_rows, self.n_cols, 3)}")
        self.target_pattern = checkerboard_pattern(self.n_rows, self.n_cols) if target_pattern is None \
            else np.array(target_pattern, dtype=np.float64)
        cfg = _lib.StgArrayConfig()
        cfg.rows, cfg.cols, cfg.action_mode = self.n_rows, self.n_cols, ACTION_MODES[action_mode]
        cfg.include_coupling, cfg.max_steps, cfg.obs_mode = int(bool(include_coupling)), int(max_steps), OBS_MODES[observation_mode]
        cfg.max_current, cfg.max_duration = float(max_current), float(max_duration)
        cfg.success_threshold, cfg.energy_penalty_weight = float(success_threshold), float(energy_penalty_weight)
        cfg.temperature = float(temperature)
        self.cfg = cfg
        self._rng, _ = _np_random(seed)
        factory_fn = HipArrayBackend if backend is None else backend
        self.backend = factory_fn(self.num_envs, cfg, flatten_params(self.device), self.coupling_matrix, device_index, env_id0)
        self._needs_reset = True

    def _soa(self, pattern):
        """[N, rows, cols, 3] or [rows, cols, 3] -> component-major [n_dev*3, N]."""
        if pattern is None:
            return None
        t = torch.as_tensor(np.asarray(pattern, dtype=np.float64) if not torch.is_tensor(pattern) else pattern).to(torch.float64)
        if t.dim() == 3:
            t = t.unsqueeze(0).expand(self.num_envs, *t.shape)
        if tuple(t.shape) != (self.num_envs, self.n_rows, self.n_cols, 3):
            raise ValueError(f"expected [N,{self.n_rows},{self.n_cols},3] or [{self.n_rows},{self.n_cols},3]")
        return t.reshape(self.num_envs, self.n_devices * 3).t().contiguous()

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        options = options or {}
        if seed is not None:
            self._rng, _ = _np_random(seed)
        dev_seed = int(self._rng.integers(0, 2**63 - 1))
        target = options.get("target_pattern")
        if target is None and self._needs_reset:
            target = self.target_pattern
        mask = options.get("mask")
        obs = self.backend.reset(None if mask is None else torch.as_tensor(mask).to(torch.uint8),
                                 self._soa(options.get("initial_pattern")), self._soa(target), dev_seed)
        self._needs_reset = False
        return self._shape_obs(obs), {}

    def _shape_obs(self, obs):
        """[obs_dim, N] kernel output -> what the observation mode promises."""
        if self.observation_mode != "dict":
            return obs.t()
        # array_env.py:569-578: the vector observation's pattern/target/similarity entries, plus the raw step budget and
        # energy (not the vector mode's normalised ones) from the device state
        n3 = self.n_devices * 3
        st = self.backend.get_state()
        shape = (self.num_envs, self.n_rows, self.n_cols, 3)
        return {"current_pattern": obs[:n3].t().reshape(shape), "target_pattern": obs[n3:2 * n3].t().reshape(shape),
                "pattern_similarity": obs[2 * n3:2 * n3 + 1].t(),
                "steps_remaining": (int(self.cfg.max_steps) - st["step_count"].to(torch.int64)).unsqueeze(1),
                "total_energy": st["total_energy"].to(torch.float32).unsqueeze(1)}

    def step(self, actions):
        if self._needs_reset:
            raise RuntimeError("Environment must be reset before calling step")
        obs, rew, rew64, term, trunc = self.backend.step(torch.as_tensor(actions).t())
        return self._shape_obs(obs), rew, term.bool(), trunc.bool(), {"reward_f64": rew64, "energy": self.backend.energy}

    def get_state(self):
        return self.backend.get_state()

    def close(self):
        self.backend.close()


class SpinTorqueArrayEnv(_EnvBase):
    """Drop-in for spin_torque_gym.envs.SpinTorqueArrayEnv (array_env.py:21-755) on the GPU path, N = 1."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 10}

    def __init__(self, array_size: Tuple[int, int] = (4, 4), device_type: str = "stt_mram",
                 device_params: Optional[Dict[str, Any]] = None, target_pattern: Optional[np.ndarray] = None,
                 max_steps: int = 200, max_current: float = 2e6, max_duration: float = 5e-9, temperature: float = 300.0,
                 include_thermal_fluctuations: bool = True, include_coupling: bool = True, coupling_strength: float = 0.1,
                 coupling_type: str = "dipolar", reward_components=None, action_mode: str = "individual",
                 observation_mode: str = "array", success_threshold: float = 0.9, energy_penalty_weight: float = 0.1,
                 render_mode: Optional[str] = None, seed: Optional[int] = None, device_index: int = 0, backend=None):
        if reward_components is not None:
            raise NotImplementedError("custom reward callables stay on the reference; the GPU path implements the default reward")
        self._vec = SpinTorqueArrayVecEnv(1, array_size, device_type, device_params, target_pattern, max_steps, max_current,
                                          max_duration, temperature, include_coupling, coupling_strength, coupling_type,
                                          action_mode, observation_mode, success_threshold, energy_penalty_weight, seed,
                                          device_index, 0, backend)
        v = self._vec
        self.array_size, self.n_rows, self.n_cols, self.n_devices = v.array_size, v.n_rows, v.n_cols, v.n_devices
        self.device_type, self.action_mode, self.observation_mode = device_type, action_mode, observation_mode
        self.max_steps, self.max_current, self.max_duration = max_steps, max_current, max_duration
        self.temperature, self.include_thermal, self.include_coupling = temperature, include_thermal_fluctuations, include_coupling
        self.coupling_strength, self.coupling_type = coupling_strength, coupling_type
        self.success_threshold, self.energy_penalty_weight, self.render_mode = success_threshold, energy_penalty_weight, render_mode
        self.devices = [v.device] * self.n_devices
        if include_coupling:
            self.coupling_matrix = v.coupling_matrix
        self.target_pattern = v.target_pattern.copy()
        hi = {"individual": self.n_devices - 1, "row": self.n_rows - 1, "column": self.n_cols - 1}
        if action_mode == "global":
            self.action_space = _box([-max_current, 0.0], [max_current, max_duration], dtype=np.float32)
        else:
            self.action_space = _box([0.0, -max_current, 0.0], [hi[action_mode], max_current, max_duration], dtype=np.float32)
        if observation_mode == "dict":                                              # array_env.py:273-287
            pat = (self.n_rows, self.n_cols, 3)
            fields = {"current_pattern": _box(-1.0, 1.0, shape=pat), "target_pattern": _box(-1.0, 1.0, shape=pat),
                      "pattern_similarity": _box(0.0, 1.0, shape=(1,)),
                      "steps_remaining": _box(0, max_steps, shape=(1,), dtype=np.int64),
                      "total_energy": _box(0.0, np.inf, shape=(1,))}
            self.observation_space = _spaces.Dict(fields) if _spaces is not None else fields
        else:
            shape = (self.n_rows, self.n_cols, 6) if observation_mode == "array" else (self.n_devices * 6 + 4,)
            self.observation_space = _box(-1.0 if observation_mode == "array" else -np.inf,
                                          1.0 if observation_mode == "array" else np.inf, shape=shape, dtype=np.float32)
        self._np_random, _ = _np_random(seed)
        self.current_pattern = None
        self.step_count, self.total_energy = 0, 0.0
        self.episode_history: List[Dict[str, Any]] = []
        self.renderer = None                  # the persistent figure of render('human') (array_env.py:149-151, 608-620)
        if render_mode == "human":
            from .render import HumanFigure
            self.renderer = HumanFigure("array")

    # -- rendering (array_env.py:594-708): host-side matplotlib, optional (spin_torque_gym_amd/render.py) --------------
    def render(self, mode: Optional[str] = None):
        """'human': m_z maps of the current and the target pattern, similarity (with the success threshold) and energy per step
        from `episode_history`, one persistent figure, returns None; 'rgb_array': the two maps as uint8 [H, W, 3]."""
        mode = self.render_mode if mode is None else mode
        if mode is None:
            return None
        if mode not in ("human", "rgb_array"):
            raise ValueError(f"Unsupported render mode: {mode}")
        if self.current_pattern is None:
            raise RuntimeError("Environment must be reset before calling render")
        from . import render as _render
        if mode == "rgb_array":
            return _render.array_rgb(self, self._similarity())
        if self.renderer is None:
            self.renderer = _render.HumanFigure("array")
        self.renderer.draw(self)
        return None

    def seed(self, seed: Optional[int] = None):
        self._np_random, s = _np_random(seed)
        return [s]

    def _shape_obs(self, obs):
        if self.observation_mode == "dict":                                         # array_env.py:569-578
            return {"current_pattern": self.current_pattern.astype(np.float32),
                    "target_pattern": self.target_pattern.astype(np.float32),
                    "pattern_similarity": np.array([self._similarity()], dtype=np.float32),
                    "steps_remaining": np.array([self.max_steps - self.step_count], dtype=int),
                    "total_energy": np.array([self.total_energy], dtype=np.float32)}
        o = obs[0].cpu().numpy().copy()
        return o.reshape(self.n_rows, self.n_cols, 6) if self.observation_mode == "array" else o

    def _pull(self):
        st = self._vec.get_state()
        self.current_pattern = st["pattern"][:, 0].cpu().numpy().reshape(self.n_rows, self.n_cols, 3).copy()
        self.total_energy = float(st["total_energy"][0])
        self.step_count = int(st["step_count"][0])

    def _similarity(self):
        return float(np.mean([np.dot(self.current_pattern[i, j], self.target_pattern[i, j])
                              for i in range(self.n_rows) for j in range(self.n_cols)]))

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        if seed is not None:
            self._np_random, _ = _np_random(seed)
        options = options or {}
        self.episode_history = []
        if "initial_pattern" in options:
            init = np.array(options["initial_pattern"], dtype=np.float64)
        else:     # array_env.py:350-356: one normal(0,1,3) draw per device, row-major, from the env's PCG64
            init = np.zeros((self.n_rows, self.n_cols, 3))
            for i in range(self.n_rows):
                for j in range(self.n_cols):
                    m = self._np_random.normal(0, 1, 3)
                    init[i, j] = m / np.linalg.norm(m)
        if "target_pattern" in options:
            self.target_pattern = np.array(options["target_pattern"], dtype=np.float64)
        obs, _ = self._vec.reset(options={"initial_pattern": init, "target_pattern": self.target_pattern})
        self._pull()
        return self._shape_obs(obs), self._get_info()

    def step(self, action):
        if self.current_pattern is None:
            raise RuntimeError("Environment must be reset before calling step")
        a = np.asarray(action, dtype=np.float32)
        if self.action_mode != "global" and np.isnan(a[0]):
            raise ValueError("cannot convert float NaN to integer")          # int(np.clip(nan, ...)), array_env.py:429
        prev_similarity = self._similarity()
        obs, rew, term, trunc, info_t = self._vec.step(torch.from_numpy(np.ascontiguousarray(a)).unsqueeze(0))
        self._pull()
        reward = float(info_t["reward_f64"][0])
        energy = float(info_t["energy"][0])
        sim = self._similarity()
        J = float(np.clip(float(a[1]) if len(a) > 1 else 0.0, -self.max_current, self.max_current))
        T = float(np.clip(float(a[2]) if len(a) > 2 else 1e-9, 1e-12, self.max_duration))
        if self.action_mode == "individual":
            affected = [int(np.clip(a[0], 0, self.n_devices - 1))]
        elif self.action_mode == "row":
            r = int(np.clip(a[0], 0, self.n_rows - 1))
            affected = list(range(r * self.n_cols, (r + 1) * self.n_cols))
        elif self.action_mode == "column":
            affected = list(range(int(np.clip(a[0], 0, self.n_cols - 1)), self.n_devices, self.n_cols))
        else:
            affected = list(range(self.n_devices))
        self.episode_history.append({"step": self.step_count, "action": a.copy(), "pattern": self.current_pattern.copy(),
                                     "reward": reward, "energy": energy, "similarity": sim})
        info = self._get_info()
        info.update({"energy_consumed": energy, "affected_devices": affected, "current_density": J, "pulse_duration": T,
                     "is_success": bool(term[0]), "step_energy": energy, "pattern_improvement": sim - prev_similarity,
                     "pattern_similarity": sim})
        return self._shape_obs(obs), reward, bool(term[0]), bool(trunc[0]), info

    def _get_info(self):
        sim = self._similarity()
        return {"step_count": self.step_count, "total_energy": self.total_energy, "pattern_similarity": sim,
                "is_success": sim >= self.success_threshold, "array_size": self.array_size,
                "device_type": self.device_type, "episode_history": self.episode_history.copy()}

    def set_target_pattern(self, pattern: np.ndarray):
        """array_env.py:735-739.  On the GPU path the new target reaches the device at the next reset()."""
        if np.shape(pattern) != (self.n_rows, self.n_cols, 3):
            raise ValueError(f"Pattern shape must be {(self.n_rows, self.n_cols, 3)}")
        self.target_pattern = np.array(pattern, dtype=np.float64)

    def get_array_info(self):
        return {"array_size": self.array_size, "n_devices": self.n_devices, "device_type": self.device_type,
                "action_mode": self.action_mode, "coupling_enabled": self.include_coupling,
                "coupling_type": self.coupling_type, "coupling_strength": self.coupling_strength}

    def close(self):
        if self.renderer is not None:
            self.renderer.close()
            self.renderer = None
        self._vec.close()
