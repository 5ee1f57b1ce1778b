"""HipBackend: one `stg_ctx` on one MI355X, driven through the C-ABI.

PyTorch-ROCm is used only as the device-memory container (tensors) and for the stream the kernels are
enqueued on; every computation happens in libspintorque_hip.so.  There is no CPU implementation here:
constructing a HipBackend without the built library or without a visible GPU raises.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib


@dataclass
class EnvConfig:
    """SpinTorqueEnv.__init__ keyword arguments that reach the step path (spin_torque_env.py:36-53) plus the
    solver constructor arguments (llgs_solver.py:24-31, simple_solver.py:24-31)."""
    solver: str = "rk4"                       # 'rk4' | 'euler' (SimpleLLGSSolver) | 'rk45' (LLGSSolver)
    include_thermal_fluctuations: bool = True
    temperature: float = 300.0
    gamma: float = 2.21e5
    max_step: float = 1e-12
    rtol: float = 1e-6
    atol: float = 1e-9
    max_steps: int = 100
    max_current: float = 2e6
    max_duration: float = 5e-9
    success_threshold: float = 0.9
    energy_penalty_weight: float = 0.1
    target_states: Sequence[Sequence[float]] = field(default_factory=lambda: [[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]])
    seed: int = 0
    max_attempts: int = 200_000               # RK45 attempts per solve before giving up (a 5 ns pulse needs ~10 000)
    skip_done: bool = False
    torque_model: str = "reference"           # 'reference' (the env's type-agnostic RHS) | 'device' (opt-in)
    lane_sort: Optional[bool] = None          # duration-sorted lane schedule: None = automatic, True/False = force
    wave_spec: Optional[bool] = None          # producer/consumer wavefront pairs (thermal): None = automatic
    noise_model: str = "white"                # 'white' (reference solvers) | 'ou' (ThermalFluctuations, fixed-step only)
    correlation_time: float = 1e-12           # ThermalFluctuations.correlation_time, for noise_model='ou'
    out_layout: str = "soa"                   # 'soa': obs [12,N] + reward/terminated/truncated arrays; 'records': one
                                              # [N,56]-byte record array (what the multi-GPU gather moves, copy-free)
    lane_refill: Optional[int] = None         # RK45 throughput launches: envs per lane (on average) of the lane-refill kernel; None =
                                              # automatic (the thresholds are in include/spintorque_hip.h, cfg.lane_refill),
                                              # False/0 = never, n >= 2 = force (results are bit-identical)
    diagnostics: bool = False                 # also write the step's fp64 reward, per-step energy and status bytes into separate
                                              # arrays (the C-ABI's optional outputs).  Off: those pointers are NULL, a step writes the
                                              # RL-facing outputs only (the status still rides in byte 54 of each record with
                                              # out_layout='records').  The terminal observations of same-step auto-reset are RL-facing
                                              # data, not a diagnostic: `final_obs` is written whenever auto-reset is on.

    def to_abi(self) -> "_lib.StgConfig":
        if self.solver not in _lib.SOLVERS:
            raise ValueError(f"Unknown solver '{self.solver}' (expected one of {list(_lib.SOLVERS)})")
        t = np.asarray(self.target_states, dtype=np.float64).reshape(-1, 3)
        if not 1 <= len(t) <= _lib.STG_MAX_TARGETS:
            raise ValueError(f"between 1 and {_lib.STG_MAX_TARGETS} target states are supported")
        c = _lib.StgConfig()
        c.solver = _lib.SOLVERS[self.solver]
        c.thermal = int(bool(self.include_thermal_fluctuations))
        c.temperature, c.gamma, c.max_step = float(self.temperature), float(self.gamma), float(self.max_step)
        c.rtol, c.atol = float(self.rtol), float(self.atol)
        c.max_steps, c.n_targets = int(self.max_steps), len(t)
        c.max_current, c.max_duration = float(self.max_current), float(self.max_duration)
        c.success_threshold, c.energy_penalty_weight = float(self.success_threshold), float(self.energy_penalty_weight)
        for i, row in enumerate(t):
            for j in range(3):
                c.targets[i][j] = float(row[j])
        c.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        c.max_attempts = int(self.max_attempts)
        c.skip_done = int(bool(self.skip_done))
        if self.torque_model not in ("reference", "device"):
            raise ValueError("torque_model must be 'reference' or 'device'")
        c.torque_model = int(self.torque_model == "device")
        c.lane_sort = 0 if self.lane_sort is None else (1 if self.lane_sort else -1)
        c.wave_spec = 0 if self.wave_spec is None else (1 if self.wave_spec else -1)
        if self.noise_model not in ("white", "ou"):
            raise ValueError("noise_model must be 'white' or 'ou'")
        c.noise_model = int(self.noise_model == "ou")
        c.noise_corr_time = float(self.correlation_time)
        if self.out_layout not in _lib.OUT_LAYOUTS:
            raise ValueError("out_layout must be 'soa' or 'records'")
        c.out_layout = _lib.OUT_LAYOUTS[self.out_layout]
        if self.lane_refill is None:
            c.lane_refill = 0
        elif not self.lane_refill:
            c.lane_refill = -1
        else:
            if int(self.lane_refill) < 2:
                raise ValueError("lane_refill must be None (automatic), False / 0 (never) or an integer >= 2 (envs per lane)")
            c.lane_refill = int(self.lane_refill)
        return c


PACKED_BYTES_PER_ENV = 54   # obs 12 x f32 + reward f32 + terminated u8 + truncated u8


def packed_step_buffer(n: int, device):
    """One contiguous uint8 buffer holding a step's RL-facing outputs, and typed views into it:
    [0, 48n) obs f32[12][n] | [48n, 52n) reward f32[n] | [52n, 53n) terminated u8[n] | [53n, 54n) truncated u8[n]."""
    buf = torch.zeros(PACKED_BYTES_PER_ENV * n, dtype=torch.uint8, device=device)
    return buf, unpack_step_buffer(buf, n)


def unpack_step_buffer(buf: torch.Tensor, n: int):
    obs = buf[: 48 * n].view(torch.float32).view(12, n)
    reward = buf[48 * n: 52 * n].view(torch.float32)
    return obs, reward, buf[52 * n: 53 * n], buf[53 * n: 54 * n]


RECORD_BYTES = _lib.RECORD_BYTES   # f32 obs[12] | f32 reward | u8 terminated | u8 truncated | u8 status | u8 0


def record_views(buf: torch.Tensor):
    """Typed views of a record array `buf` uint8[..., n, 56] (STG_OUT_RECORDS), no copies:
    obs f32[..., n, 12] (row stride 14 floats -- Gym's [N, 12] orientation), reward f32[..., n], terminated / truncated /
    status u8[..., n]."""
    f = buf.view(torch.float32)                       # [..., n, 14]
    return f[..., :12], f[..., 12], buf[..., 52], buf[..., 53], buf[..., 54]


def alloc_step_outputs(n: int, device, layout: str):
    """The RL-facing outputs of one step in the given layout: (container, obs [12,n] view, reward, terminated, truncated).
    'soa': the 54 B/env packed byte buffer (component-major obs); 'records': uint8 [n,56], obs = the transposed view."""
    if layout == "records":
        rec = torch.zeros((n, RECORD_BYTES), dtype=torch.uint8, device=device)
        obs, reward, term, trunc, _ = record_views(rec)
        return rec, obs.t(), reward, term, trunc
    buf, (obs, reward, term, trunc) = packed_step_buffer(n, device)
    return buf, obs, reward, term, trunc


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipBackend:
    """Owns one stg_ctx.  All tensors are on ``cuda:<device_index>``; per-env arrays are component-major
    ([3,N], [12,N], [2,N]) as the C-ABI defines them."""

    def __init__(self, n_envs: int, cfg: EnvConfig, device_index: int = 0, env_id0: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs a visible MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the step path")
        self.n = int(n_envs)
        self.cfg = cfg
        self.device = torch.device("cuda", device_index)
        self.env_id0 = int(env_id0)
        self._ctx = C.c_void_p()
        abi = cfg.to_abi()
        _lib.check(self.lib.stg_create(C.byref(self._ctx), device_index, self.n, self.env_id0, C.byref(abi)))
        self._cls = None
        n = self.n
        dev = self.device
        # (obs, reward, terminated, truncated) live in ONE buffer; the tensors below are views of it.  'soa': 54 B/env,
        # component-major obs [12,N]; 'records': [N,56] bytes, env-major -- a shard is one contiguous block, so the
        # multi-GPU path moves a step's results with a single all-gather straight into the global record array
        #  and `obs` is a strided [12,N] view (its .t() is Gym's [N,12]).
        self.records_layout = cfg.out_layout == "records"
        self.packed, self.obs, self.reward, self.terminated, self.truncated = alloc_step_outputs(n, dev, cfg.out_layout)
        self.diagnostics = bool(getattr(cfg, "diagnostics", False))
        if self.diagnostics:
            self.reward64 = torch.empty(n, dtype=torch.float64, device=dev)
            self.energy = torch.empty(n, dtype=torch.float64, device=dev)
            self.status = torch.empty(n, dtype=torch.uint8, device=dev)
        else:
            # the hot path writes nothing but the RL-facing outputs; in the records layout the status byte of each record
            # is still there (a view), in the SoA layout there is no status without diagnostics
            self.reward64 = self.energy = None
            self.status = record_views(self.packed)[4] if self.records_layout else None
        self.final_obs = None                 # [12,N] (a view of [N,12] in the records layout), allocated on the first auto-resetting step
        self._final_buf = None

    # -- plumbing -------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype, shape=None):
        if t is None:
            return None
        t = torch.as_tensor(t)
        t = t.to(device=self.device, dtype=dtype).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def close(self):
        if self._ctx:
            torch.cuda.synchronize(self.device)
            self.lib.stg_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameter surface ------------------------------------------------------------------------
    def set_params(self, table: List["_lib.StgDeviceParams"], cls=None):
        arr = (_lib.StgDeviceParams * len(table))(*table)
        self._cls = self._dev(cls, torch.uint8, (self.n,)) if len(table) > 1 else None
        if len(table) > 1 and self._cls is None:
            raise ValueError("a class index per env is required with more than one device class")
        if self._cls is not None and int(self._cls.max()) >= len(table):
            raise ValueError("class index out of range")
        _lib.check(self.lib.stg_set_params(self._ctx, arr, len(table), _ptr(self._cls)))
        self.n_classes = len(table)

    def set_params_per_env(self, block, dev_type, valid):
        """Per-env parameters: block float64 [STG_NPARAM, N] (rows = the double fields of stg_device_params in
        declaration order, see devices.PARAM_ROWS / per_env_param_block), dev_type uint8 [N], valid uint8 [N]."""
        block = self._dev(block, torch.float64, (_lib.STG_NPARAM, self.n))
        dev_type = self._dev(dev_type, torch.uint8, (self.n,))
        valid = self._dev(valid, torch.uint8, (self.n,))
        torch.cuda.synchronize(self.device)
        _lib.check(self.lib.stg_set_params_per_env(self._ctx, _ptr(block), _ptr(dev_type), _ptr(valid)))
        self._cls = None
        self.n_classes = 0

    def thermal_strength(self, cls=0) -> float:
        out = C.c_double()
        _lib.check(self.lib.stg_thermal_strength(self._ctx, cls, C.byref(out)))
        return out.value

    # -- env level ----------------------------------------------------------------------------------
    def reset(self, mask=None, init_m=None, target=None, seed=0):
        mask = self._dev(mask, torch.uint8, (self.n,))
        init_m = self._dev(init_m, torch.float64, (3, self.n))
        target = self._dev(target, torch.float64, (3, self.n))
        _lib.check(self.lib.stg_reset(self._ctx, _ptr(mask), _ptr(init_m), _ptr(target),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF,
                                      _ptr(self.packed if self.records_layout else self.obs), self._stream()))
        self._keep = (mask, init_m, target)      # keep inputs alive until the stream has consumed them
        return self.obs

    def step(self, actions, autoreset=False, out: Optional[torch.Tensor] = None):
        """actions: [2,N] float32 or float64 tensor (row 0 current density, row 1 duration).  One kernel launch; the
        RL-facing outputs land in `self.packed`.  autoreset (same-step): an env whose episode ends ON THIS step reports
        this step's reward / terminated / truncated, is reset on the device inside the same launch, and its `obs` row
        already holds the new episode's first observation; the terminal observation goes to `self.final_obs` (always, with or
        without `diagnostics`: a learner that bootstraps at truncation needs it; rows of other envs keep their last content).
        out ('records' layout only): another uint8 [N,56] record array to write this step's outputs into instead of
        `self.packed` (callers that double-buffer, e.g. the pipelined multi-GPU gather); the returned views are of `out`."""
        a = torch.as_tensor(actions)
        f64 = a.dtype == torch.float64
        a = self._dev(a, torch.float64 if f64 else torch.float32, (2, self.n))
        if autoreset and self.final_obs is None:
            if self.records_layout:           # env-major like the records: one 48-byte block per env
                self._final_buf = torch.zeros((self.n, 12), dtype=torch.float32, device=self.device)
                self.final_obs = self._final_buf.t()
            else:
                self._final_buf = self.final_obs = torch.zeros((12, self.n), dtype=torch.float32, device=self.device)
        if self.records_layout:
            rec = self.packed if out is None else out
            if rec.dtype != torch.uint8 or tuple(rec.shape) != (self.n, RECORD_BYTES) or not rec.is_contiguous() or rec.device != self.device:
                raise ValueError(f"out must be a contiguous uint8 [{self.n}, {RECORD_BYTES}] tensor on {self.device}")
            diag = self.diagnostics
            _lib.check(self.lib.stg_step_many(self._ctx, 1, _ptr(a), int(f64), 1, int(bool(autoreset)), _ptr(rec),
                                              _ptr(self._final_buf) if autoreset else None, None, _ptr(self.reward64),
                                              _ptr(self.energy), None, None, _ptr(self.status) if diag else None, self._stream()))
            self._keep = (a,)
            if out is None:
                return self.obs, self.reward, self.reward64, self.terminated, self.truncated, self.status
            obs, reward, term, trunc, st = record_views(rec)
            return obs.t(), reward, self.reward64, term, trunc, (self.status if diag else st)
        if out is not None:
            raise ValueError("out= needs out_layout='records'")
        _lib.check(self.lib.stg_step_many(self._ctx, 1, _ptr(a), int(f64), 1, int(bool(autoreset)), _ptr(self.obs),
                                          _ptr(self.final_obs) if autoreset else None,
                                          _ptr(self.reward), _ptr(self.reward64), _ptr(self.energy),
                                          _ptr(self.terminated), _ptr(self.truncated), _ptr(self.status),
                                          self._stream()))
        self._keep = (a,)
        return self.obs, self.reward, self.reward64, self.terminated, self.truncated, self.status

    def step_many(self, actions, out_every=True, autoreset=False):
        """actions: [K,2,N]; returns tensors with a leading K (out_every) or 1 dimension."""
        a = torch.as_tensor(actions)
        f64 = a.dtype == torch.float64
        K = int(a.shape[0])
        a = self._dev(a, torch.float64 if f64 else torch.float32, (K, 2, self.n))
        ko = K if out_every else 1
        n, dev = self.n, self.device
        if self.records_layout:
            rec = torch.empty((ko, n, RECORD_BYTES), dtype=torch.uint8, device=dev)
            o, reward, term, trunc, _ = record_views(rec)
            obs, out_ptr = o.transpose(1, 2), rec             # [ko,12,n] view
        else:
            obs = torch.empty((ko, 12, n), dtype=torch.float32, device=dev)
            reward = torch.empty((ko, n), dtype=torch.float32, device=dev)
            term = torch.empty((ko, n), dtype=torch.uint8, device=dev)
            trunc = torch.empty((ko, n), dtype=torch.uint8, device=dev)
            out_ptr = obs
        diag = self.diagnostics
        reward64 = torch.empty((ko, n), dtype=torch.float64, device=dev) if diag else None
        status = torch.empty((ko, n), dtype=torch.uint8, device=dev) if diag else (record_views(rec)[4] if self.records_layout else None)
        self.energy_many = torch.empty((ko, n), dtype=torch.float64, device=dev) if diag else None
        fbuf = None
        self.final_obs_many = None
        if autoreset:
            if self.records_layout:
                fbuf = torch.zeros((ko, n, 12), dtype=torch.float32, device=dev)
                self.final_obs_many = fbuf.transpose(1, 2)
            else:
                fbuf = self.final_obs_many = torch.zeros((ko, 12, n), dtype=torch.float32, device=dev)
        _lib.check(self.lib.stg_step_many(self._ctx, K, _ptr(a), int(f64), int(bool(out_every)), int(bool(autoreset)),
                                          _ptr(out_ptr), _ptr(fbuf), None if self.records_layout else _ptr(reward),
                                          _ptr(reward64), _ptr(self.energy_many), None if self.records_layout else _ptr(term),
                                          None if self.records_layout else _ptr(trunc), _ptr(status) if diag else None, self._stream()))
        self._keep = (a,)
        return obs, reward, reward64, term, trunc, status

    def get_state(self):
        n, dev = self.n, self.device
        st = dict(m=torch.empty((3, n), dtype=torch.float64, device=dev),
                  target=torch.empty((3, n), dtype=torch.float64, device=dev),
                  total_energy=torch.empty(n, dtype=torch.float64, device=dev),
                  step_count=torch.empty(n, dtype=torch.int32, device=dev),
                  rng_step=torch.empty(n, dtype=torch.int32, device=dev),
                  done=torch.empty(n, dtype=torch.uint8, device=dev))
        _lib.check(self.lib.stg_get_state(self._ctx, _ptr(st["m"]), _ptr(st["target"]), _ptr(st["total_energy"]),
                                          _ptr(st["step_count"]), _ptr(st["rng_step"]), _ptr(st["done"]),
                                          self._stream()))
        return st

    def set_state(self, st):
        n = self.n
        m = self._dev(st.get("m"), torch.float64, (3, n))
        tg = self._dev(st.get("target"), torch.float64, (3, n))
        e = self._dev(st.get("total_energy"), torch.float64, (n,))
        sc = self._dev(st.get("step_count"), torch.int32, (n,))
        rs = self._dev(st.get("rng_step"), torch.int32, (n,))
        dn = self._dev(st.get("done"), torch.uint8, (n,))
        _lib.check(self.lib.stg_set_state(self._ctx, _ptr(m), _ptr(tg), _ptr(e), _ptr(sc), _ptr(rs), _ptr(dn),
                                          self._stream()))
        self._keep = (m, tg, e, sc, rs, dn)

    # -- solver level -------------------------------------------------------------------------------
    def solve(self, m0, J, T, env_step=0, traj_cap=0, want_energy=False):
        n, dev = self.n, self.device
        m0 = self._dev(m0, torch.float64, (3, n))
        J = self._dev(J, torch.float64, (n,))
        T = self._dev(T, torch.float64, (n,))
        mf = torch.empty((3, n), dtype=torch.float64, device=dev)
        npts = torch.empty(n, dtype=torch.int32, device=dev)
        succ = torch.empty(n, dtype=torch.uint8, device=dev)
        out = dict(m_final=mf, n_points=npts, success=succ)
        if traj_cap > 0:
            t = torch.zeros((traj_cap, n), dtype=torch.float64, device=dev)
            m = torch.zeros((traj_cap, 3, n), dtype=torch.float64, device=dev)
            # energy and torques are LLGSSolver's by-products (llgs_solver.py:154-172), recorded together
            e = torch.zeros((traj_cap, n), dtype=torch.float64, device=dev) if want_energy else None
            tq = torch.zeros((traj_cap, n), dtype=torch.float64, device=dev) if want_energy else None
            _lib.check(self.lib.stg_solve_traj(self._ctx, _ptr(m0), _ptr(J), _ptr(T), int(env_step), int(traj_cap),
                                               _ptr(t), _ptr(m), _ptr(e), _ptr(tq), _ptr(mf), _ptr(npts), _ptr(succ),
                                               self._stream()))
            out.update(t=t, m=m, energy=e, torques=tq)
        else:
            _lib.check(self.lib.stg_solve(self._ctx, _ptr(m0), _ptr(J), _ptr(T), int(env_step), _ptr(mf),
                                          _ptr(npts), _ptr(succ), self._stream()))
        self._keep = (m0, J, T)
        return out

    def device_terms(self, m, J, volt):
        """Device-class formulas per env: (tau_dl [3,N], tau_fl [3,N], k_eff [N]) for m [3,N], J [N], volt [N]."""
        n, dev = self.n, self.device
        m = self._dev(m, torch.float64, (3, n))
        J = self._dev(J, torch.float64, (n,))
        volt = self._dev(volt, torch.float64, (n,))
        dl = torch.empty((3, n), dtype=torch.float64, device=dev)
        fl = torch.empty((3, n), dtype=torch.float64, device=dev)
        ke = torch.empty(n, dtype=torch.float64, device=dev)
        _lib.check(self.lib.stg_device_terms(self._ctx, _ptr(m), _ptr(J), _ptr(volt), _ptr(dl), _ptr(fl), _ptr(ke),
                                             self._stream()))
        self._keep = (m, J, volt)
        return dl, fl, ke

    def counters(self, reset=False):
        """On-device metrics: dict(env_steps, work_units, noop_steps).  Synchronises the device."""
        out = (C.c_uint64 * 4)()
        _lib.check(self.lib.stg_get_counters(self._ctx, out, int(bool(reset))))
        return {"env_steps": int(out[0]), "work_units": int(out[1]), "noop_steps": int(out[3])}

    PLACEMENT_CAP = 4096
    PLACEMENT_ENTRY = 4                   # STG_PLACEMENT_WORDS_PER_WAVE

    def placement(self, launches_back=0, raw=False):
        """Where the dispatcher put the wavefronts of a recent step launch and when each ran (stg_get_placement; 0 = the latest, up to
        31 back): dict(workgroups, waves_per_workgroup, waves_recorded, simds_used, integrating_per_simd (histogram {count: SIMDs}),
        simd_double_booked, span_us, simd_busy_frac, last_simd_alone_frac).
        `simd_double_booked` = SIMDs that held two or more INTEGRATING wavefronts of the launch -- for the launches scheduled for one
        integrating wavefront per SIMD (up to 65 536 envs; the wave-specialised pairs) it should be 0, and every such SIMD costs the
        launch the time of its two wavefronts back to back.  From the start / retire times: `span_us` = first start to last retire;
        `simd_busy_frac` = mean over the 1024 SIMDs of the share of that span during which the SIMD held at least one integrating
        wavefront; `last_simd_alone_frac` = share of the span left once 90 % of the SIMDs in use have retired their last integrating
        wavefront (the tail a makespan-bound launch ends with).  raw=True adds the arrays (`where`, `producer`, `t0_us`, `t1_us`, `work`, `slot` per recorded wavefront).
        Synchronises the device."""
        words = self.PLACEMENT_CAP * self.PLACEMENT_ENTRY
        out = (C.c_uint32 * words)()
        nwg, wpw = C.c_int32(), C.c_int32()
        n = self.lib.stg_get_placement(self._ctx, int(launches_back), out, words, C.byref(nwg), C.byref(wpw))
        if n < 0:
            _lib.check(n)
        e = np.frombuffer(out, dtype=np.uint32, count=n * self.PLACEMENT_ENTRY).reshape(n, self.PLACEMENT_ENTRY).copy()
        e = e[(e[:, 0] >> 31) == 1]
        w = e[:, 0]
        simd = (w & 0xFFFF) >> 4 & 3
        where = ((w >> 16 & 0xF).astype(np.int64) << 12) | ((w >> 8 & 0xFF).astype(np.int64) << 2) | simd      # (xcc, se/sh/cu, simd)
        prod = (w >> 20 & 1) == 1
        integ = where[~prod]
        _, cnt = np.unique(integ, return_counts=True)
        hist = {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))}
        res = {"workgroups": int(nwg.value), "waves_per_workgroup": int(wpw.value), "waves_recorded": int(len(w)),
               "simds_used": int(len(np.unique(where))), "integrating_per_simd": hist,
               "simd_double_booked": int((cnt >= 2).sum())}
        # timeline (100 MHz ticks, 32-bit wrap-safe differences against the earliest start)
        ran = e[:, 2] != 0
        if ran.any():
            t0 = e[ran, 1].astype(np.int64)
            base = int(t0.min()) if (int(t0.max()) - int(t0.min())) < (1 << 31) else int(t0[0])
            s0 = ((e[:, 1].astype(np.int64) - base) & 0xFFFFFFFF).astype(np.float64) * 0.01
            s1 = ((e[:, 2].astype(np.int64) - base) & 0xFFFFFFFF).astype(np.float64) * 0.01
            s0, s1 = np.where(ran, s0, 0.0), np.where(ran, s1, 0.0)
            sel = ran & ~prod
            span = float(s1[sel].max() - s0[sel].min()) if sel.any() else 0.0
            busy = {}
            for k in np.unique(where[sel]):           # union of the integrating wavefronts' intervals per SIMD
                iv = sorted(zip(s0[sel & (where == k)], s1[sel & (where == k)]))
                tot, cur_a, cur_b = 0.0, iv[0][0], iv[0][1]
                for a_, b_ in iv[1:]:
                    if a_ > cur_b:
                        tot += cur_b - cur_a
                        cur_a, cur_b = a_, b_
                    else:
                        cur_b = max(cur_b, b_)
                busy[int(k)] = (tot + cur_b - cur_a, cur_b)
            n_simd = 1024
            ends = np.sort(np.array([v[1] for v in busy.values()]))
            res["span_us"] = round(span, 2)
            res["simd_busy_frac"] = round(sum(v[0] for v in busy.values()) / (n_simd * span), 4) if span > 0 else None
            # from the moment 90 % of the SIMDs in use have retired their last integrating wavefront the launch runs on a tenth of them
            t90 = float(ends[max(int(0.9 * len(ends)) - 1, 0)])
            res["last_simd_alone_frac"] = round(max(0.0, float(ends[-1]) - t90) / span, 4) if span > 0 else None
            if raw:
                res.update(where=where, producer=prod, t0_us=s0, t1_us=s1, work=e[:, 3].astype(np.int64), slot=(w & 0xF).astype(np.int64))
        return res

    def thermal_normals(self, env_step=0, call0=0, n_calls=1):
        z = torch.empty((n_calls, 3, self.n), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.stg_thermal_normals(self._ctx, int(env_step), int(call0), int(n_calls), _ptr(z),
                                                self._stream()))
        return z
