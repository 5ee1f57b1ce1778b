"""Test-only helpers.  `OracleBackend` offers HipBackend's interface on top of the CPU oracle so that

  * `-m gpu` parity tests can run the SAME host code (SpinTorqueVecEnv / SpinTorqueEnv) once over the HIP
    library and once over the oracle and compare the outputs, and
  * the world_size-2 gloo tests can exercise sharding and the (obs, reward, done) gather without a GPU.

It lives under tests/ because only tests may touch oracle/ (the product never does).
"""
import ctypes as C

import numpy as np
import torch

import oracle

RESET_KEY_XOR = 0x9E3779B97F4A7C15


def oracle_config(cfg):
    """spin_torque_gym_amd.backend.EnvConfig -> oracle.Config"""
    return oracle.make_config(solver=cfg.solver, thermal=cfg.include_thermal_fluctuations, temperature=cfg.temperature,
                              gamma=cfg.gamma, max_step=cfg.max_step, rtol=cfg.rtol, atol=cfg.atol,
                              max_steps=cfg.max_steps, max_current=cfg.max_current, max_duration=cfg.max_duration,
                              success_threshold=cfg.success_threshold, energy_penalty_weight=cfg.energy_penalty_weight,
                              seed=cfg.seed, max_attempts=cfg.max_attempts,
                              torque_model=int(getattr(cfg, "torque_model", "reference") == "device"),
                              noise_model=int(getattr(cfg, "noise_model", "white") == "ou"),
                              noise_corr_time=getattr(cfg, "correlation_time", 1e-12))


def oracle_params(table):
    """list of _lib.StgDeviceParams -> (oracle.Params * k); the two records have the same field layout."""
    arr = (oracle.Params * len(table))()
    assert C.sizeof(oracle.Params) == C.sizeof(type(table[0]))
    for i, p in enumerate(table):
        C.memmove(C.byref(arr[i]), C.byref(p), C.sizeof(oracle.Params))
    return arr


def _philox(ctr, key):
    L = oracle.lib()
    L.stgo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.stgo_philox4x32_10.restype = None
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    L.stgo_philox4x32_10(c, k, out)
    return [int(x) for x in out]


def device_reset_draw(seed, env_id, rng_step, targets):
    """The kernels' device-side reset draw (csrc/spintorque_hip.hip: device_reset_draw) through its oracle restatement."""
    L = oracle.lib()
    L.stgo_reset_draw.restype = None
    L.stgo_reset_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    z = np.zeros(3)
    idx = C.c_int(0)
    L.stgo_reset_draw(seed & 0xFFFFFFFFFFFFFFFF, env_id, rng_step, len(targets), z.ctypes.data_as(C.POINTER(C.c_double)),
                      C.byref(idx))
    n = np.sqrt((z[0] * z[0] + z[1] * z[1]) + z[2] * z[2])
    m = np.array([0.0, 0.0, 1.0]) if n < 1e-12 else z / n
    return m, np.array(targets[idx.value], dtype=np.float64)


class OracleBackend:
    """HipBackend's interface, computed by the CPU oracle.  Tensors are CPU torch tensors, component-major."""

    def __init__(self, n_envs, cfg, device_index=0, env_id0=0):
        self.n = int(n_envs)
        self.cfg = cfg
        self.ocfg = oracle_config(cfg)
        self.env_id0 = int(env_id0)
        self.device = torch.device("cpu")
        self.states = (oracle.EnvState * self.n)()
        self.done = np.zeros(self.n, dtype=np.uint8)
        self.params = None
        self.cls = None
        n = self.n
        from spin_torque_gym_amd.backend import alloc_step_outputs
        self.records_layout = getattr(cfg, "out_layout", "soa") == "records"
        self.packed, self.obs, self.reward, self.terminated, self.truncated = alloc_step_outputs(
            n, self.device, getattr(cfg, "out_layout", "soa"))
        self.reward64 = torch.zeros(n, dtype=torch.float64)
        self.energy = torch.zeros(n, dtype=torch.float64)
        self.status = torch.zeros(n, dtype=torch.uint8)
        self.final_obs = torch.zeros((12, n), dtype=torch.float32)
        self._counters = {"env_steps": 0, "work_units": 0, "noop_steps": 0}

    def close(self):
        pass

    def counters(self, reset=False):
        c = dict(self._counters)
        if reset:
            self._counters = {"env_steps": 0, "work_units": 0, "noop_steps": 0}
        return c

    def set_params(self, table, cls=None):
        self.params = oracle_params(table)
        self.n_classes = len(table)
        self.cls = None if (cls is None or len(table) == 1) else np.ascontiguousarray(torch.as_tensor(cls).cpu().numpy(), dtype=np.uint8)

    def _p(self, i):
        return self.params[int(self.cls[i]) if self.cls is not None else 0]

    def _obs_of(self, i, out):
        o = (C.c_float * 12)()
        oracle.lib().stgo_observation.restype = None
        oracle.lib().stgo_observation.argtypes = [C.POINTER(oracle.EnvState), C.POINTER(oracle.Params),
                                                  C.POINTER(oracle.Config), C.POINTER(C.c_float)]
        oracle.lib().stgo_observation(C.byref(self.states[i]), C.byref(self._p(i)), C.byref(self.ocfg), o)
        out[:, i] = torch.tensor(list(o), dtype=torch.float32)

    def reset(self, mask=None, init_m=None, target=None, seed=0):
        mask = None if mask is None else torch.as_tensor(mask).cpu().numpy().astype(bool)
        init_m = None if init_m is None else torch.as_tensor(init_m).cpu().numpy().astype(np.float64)
        target = None if target is None else torch.as_tensor(target).cpu().numpy().astype(np.float64)
        targets = np.asarray(self.cfg.target_states, dtype=np.float64).reshape(-1, 3)
        for i in range(self.n):
            s = self.states[i]
            if mask is not None and not mask[i]:
                s.last_action[:] = [0.0, 0.0]
                self._obs_of(i, self.obs)
                continue
            m, t = device_reset_draw(int(seed) & 0xFFFFFFFFFFFFFFFF, self.env_id0 + i, s.rng_step, targets)
            if init_m is not None:
                v = init_m[:, i]
                m = v / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            if target is not None:
                v = target[:, i]
                t = v / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            s.m[:] = list(m)
            s.target[:] = list(t)
            s.total_energy = 0.0
            s.step_count = 0
            s.last_action[:] = [0.0, 0.0]
            self.done[i] = 0
            self._obs_of(i, self.obs)
        return self.obs

    def _step_once(self, a, autoreset, outs, final=None):
        obs, reward, reward64, energy, term, trunc, status = outs
        final = self.final_obs if final is None else final
        a = np.ascontiguousarray(a.T)          # [N,2]
        f64 = a.dtype == np.float64
        targets = np.asarray(self.cfg.target_states, dtype=np.float64).reshape(-1, 3)
        if f64:
            # float64 actions: the safety clamp runs in float64 (monitoring.py:304-313 on a float64 array); the
            # oracle's C entry takes float32, so apply the clamp here and pass values float32 can carry exactly
            raise NotImplementedError("OracleBackend handles float32 actions; float64 parity is tested separately")
        outs_c = oracle.env_step_batch(self.states, a.astype(np.float32), self.params, self.cls, self.ocfg,
                                       env_id0=self.env_id0)
        for i in range(self.n):
            o = outs_c[i]
            obs[:, i] = torch.tensor(list(o.obs), dtype=torch.float32)
            reward[i] = float(np.float32(o.reward))
            reward64[i] = o.reward
            energy[i] = o.energy
            term[i] = o.terminated
            trunc[i] = o.truncated
            status[i] = o.status
            self._counters["env_steps"] += 1
            self._counters["work_units"] += int(o.n_sub)
            self._counters["noop_steps"] += int(o.status == 1)
            self.done[i] = 1 if (o.terminated or o.truncated) else 0
            if autoreset and self.done[i]:
                # same-step auto-reset (csrc/spintorque_hip.hip): terminal obs -> final_obs, redraw, obs of the new episode
                s = self.states[i]
                final[:, i] = obs[:, i]
                m, t = device_reset_draw(self.cfg.seed, self.env_id0 + i, s.rng_step, targets)
                s.m[:] = list(m)
                s.target[:] = list(t)
                s.total_energy = 0.0
                s.step_count = 0
                s.last_action[:] = [0.0, 0.0]
                self.done[i] = 0
                self._obs_of(i, obs)

    def step(self, actions, autoreset=False, out=None):
        a = torch.as_tensor(actions).cpu().numpy()
        obs, reward, term, trunc = self.obs, self.reward, self.terminated, self.truncated
        if out is not None:                      # 'records' layout: the caller's record array (double buffering)
            from spin_torque_gym_amd.backend import record_views
            assert self.records_layout and tuple(out.shape) == (self.n, 56)
            o, reward, term, trunc, st_b = record_views(out)
            obs = o.t()
        self._step_once(a, autoreset, (obs, reward, self.reward64, self.energy, term, trunc, self.status))
        if self.records_layout:                  # the record's status byte
            (out if out is not None else self.packed)[:, 54] = self.status
        return obs, reward, self.reward64, term, trunc, self.status

    def step_many(self, actions, out_every=True, autoreset=False):
        a = torch.as_tensor(actions).cpu().numpy()
        K, n = a.shape[0], self.n
        ko = K if out_every else 1
        obs = torch.zeros((ko, 12, n), dtype=torch.float32)
        reward = torch.zeros((ko, n), dtype=torch.float32)
        reward64 = torch.zeros((ko, n), dtype=torch.float64)
        self.energy_many = torch.zeros((ko, n), dtype=torch.float64)
        term = torch.zeros((ko, n), dtype=torch.uint8)
        trunc = torch.zeros((ko, n), dtype=torch.uint8)
        status = torch.zeros((ko, n), dtype=torch.uint8)
        self.final_obs_many = torch.zeros((ko, 12, n), dtype=torch.float32)
        for k in range(K):
            j = k if out_every else 0
            self._step_once(a[k], autoreset, (obs[j], reward[j], reward64[j], self.energy_many[j], term[j], trunc[j], status[j]),
                            final=self.final_obs_many[j])
        return obs, reward, reward64, term, trunc, status

    def get_state(self):
        n = self.n
        m = np.array([list(self.states[i].m) for i in range(n)]).T
        t = np.array([list(self.states[i].target) for i in range(n)]).T
        return dict(m=torch.from_numpy(np.ascontiguousarray(m)), target=torch.from_numpy(np.ascontiguousarray(t)),
                    total_energy=torch.tensor([self.states[i].total_energy for i in range(n)], dtype=torch.float64),
                    step_count=torch.tensor([self.states[i].step_count for i in range(n)], dtype=torch.int32),
                    rng_step=torch.tensor([self.states[i].rng_step for i in range(n)], dtype=torch.int64).to(torch.int32),
                    done=torch.from_numpy(self.done.copy()))

    def set_state(self, st):
        for i in range(self.n):
            s = self.states[i]
            if st.get("m") is not None:
                s.m[:] = [float(x) for x in st["m"][:, i]]
            if st.get("target") is not None:
                s.target[:] = [float(x) for x in st["target"][:, i]]
            if st.get("total_energy") is not None:
                s.total_energy = float(st["total_energy"][i])
            if st.get("step_count") is not None:
                s.step_count = int(st["step_count"][i])
            if st.get("rng_step") is not None:
                s.rng_step = int(st["rng_step"][i]) & 0xFFFFFFFF
            if st.get("done") is not None:
                self.done[i] = int(st["done"][i])

    def solve(self, m0, J, T, env_step=0, traj_cap=0, want_energy=False):
        m0 = torch.as_tensor(m0).cpu().numpy().astype(np.float64)
        J = torch.as_tensor(J).cpu().numpy().astype(np.float64)
        T = torch.as_tensor(T).cpu().numpy().astype(np.float64)
        n = self.n
        mf = np.zeros((3, n))
        npts = np.zeros(n, dtype=np.int32)
        succ = np.zeros(n, dtype=np.uint8)
        tt = np.zeros((max(traj_cap, 1), n))
        tm = np.zeros((max(traj_cap, 1), 3, n))
        te = np.zeros((max(traj_cap, 1), n))
        tq = np.zeros((max(traj_cap, 1), n))
        for i in range(n):
            p = self._p(i)
            if self.cfg.solver == "rk45":
                r = oracle.llgs_solve(m0[:, i], T[i], p, self.ocfg, J[i], self.env_id0 + i, env_step, cap=max(traj_cap, 1))
                mf[:, i] = r["m_final"] if r["success"] else m0[:, i]
                npts[i] = r["n_points"] - 1
                k = min(len(r["t"]), traj_cap)
                tt[:k, i], tm[:k, :, i], te[:k, i], tq[:k, i] = r["t"][:k], r["m"][:k], r["energy"][:k], r["torques"][:k]
            else:
                r = oracle.simple_solve(m0[:, i], T[i], p, self.ocfg, J[i], self.env_id0 + i, env_step, want_traj=traj_cap > 0)
                mf[:, i] = r["m_final"]
                npts[i] = r["n_steps"]
                if traj_cap > 0 and "m" in r:
                    k = min(len(r["m"]), traj_cap)
                    tm[:k, :, i] = r["m"][:k]
                    dt = T[i] / max(r["n_steps"], 1)
                    tt[:k, i] = np.arange(k) * dt
                    if k == r["n_steps"] + 1:
                        tt[k - 1, i] = T[i]
            succ[i] = r["success"]
        out = dict(m_final=torch.from_numpy(mf), n_points=torch.from_numpy(npts), success=torch.from_numpy(succ))
        if traj_cap > 0:
            out.update(t=torch.from_numpy(tt), m=torch.from_numpy(tm), energy=torch.from_numpy(te) if want_energy else None,
                       torques=torch.from_numpy(tq) if want_energy else None)
        return out


def make_states(n, m, target):
    st = (oracle.EnvState * n)()
    for i in range(n):
        st[i].m[:] = [float(x) for x in m[i]]
        st[i].target[:] = [float(x) for x in target[i]]
    return st


def unit_rows(rng, n):
    v = rng.normal(0, 1, (n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


class OracleArrayBackend:
    """HipArrayBackend's interface on the CPU oracle (test seam of SpinTorqueArrayVecEnv)."""

    def __init__(self, n_arrays, cfg, dev_params, coupling, device_index=0, env_id0=0):
        self.n, self.cfg = int(n_arrays), cfg
        self.n_dev = cfg.rows * cfg.cols
        self.obs_dim = self.n_dev * 6 + (4 if cfg.obs_mode == 1 else 0)
        self.device = torch.device("cpu")
        self.ocfg = oracle.ArrayConfig()
        C.memmove(C.byref(self.ocfg), C.byref(cfg), C.sizeof(oracle.ArrayConfig))
        self.p = oracle_params([dev_params])[0]
        self.coupling = np.zeros((self.n_dev, self.n_dev)) if coupling is None else np.ascontiguousarray(coupling, dtype=np.float64)
        self.states = [None] * self.n
        self.target = None
        n = self.n
        self.obs = torch.zeros((self.obs_dim, n), dtype=torch.float32)
        self.reward = torch.zeros(n, dtype=torch.float32)
        self.reward64 = torch.zeros(n, dtype=torch.float64)
        self.energy = torch.zeros(n, dtype=torch.float64)
        self.terminated = torch.zeros(n, dtype=torch.uint8)
        self.truncated = torch.zeros(n, dtype=torch.uint8)

    def close(self):
        pass

    def reset(self, mask=None, init_pattern=None, target=None, seed=0):
        assert init_pattern is not None, "the oracle backend takes explicit initial patterns"
        ip = torch.as_tensor(init_pattern).numpy()
        for i in range(self.n):
            if mask is not None and not bool(mask[i]):
                continue
            tg = torch.as_tensor(target).numpy()[:, i].reshape(-1, 3) if target is not None else self.states[i].target
            self.states[i] = oracle.ArrayEnvState(ip[:, i].reshape(-1, 3), tg)
            self.obs[:, i] = torch.from_numpy(oracle.array_observation(self.states[i], self.ocfg))
        return self.obs

    def step(self, actions):
        a = torch.as_tensor(actions).numpy().astype(np.float32)
        for i in range(self.n):
            obs, r, te, tr, en = oracle.array_step(self.states[i], a[:, i], self.p, self.ocfg, self.coupling)
            self.obs[:, i] = torch.from_numpy(obs)
            self.reward[i] = float(np.float32(r)); self.reward64[i] = r; self.energy[i] = en
            self.terminated[i] = int(te); self.truncated[i] = int(tr)
        return self.obs, self.reward, self.reward64, self.terminated, self.truncated

    def get_state(self):
        return dict(pattern=torch.from_numpy(np.stack([s.pattern.reshape(-1) for s in self.states], axis=1).copy()),
                    target=torch.from_numpy(np.stack([s.target.reshape(-1) for s in self.states], axis=1).copy()),
                    total_energy=torch.tensor([s.total_energy.value for s in self.states], dtype=torch.float64),
                    step_count=torch.tensor([s.step_count.value for s in self.states], dtype=torch.int32))
