"""ctypes binding of oracle/_build/libstg_oracle.so (the C restatement in stg_oracle.c).

TEST INFRASTRUCTURE ONLY -- the checker the HIP path is compared with, and the "port" CPU
baseline bench.py reports.  Parity status: pinned by tests/golden/ (generated from the imported
Python reference by tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstg_oracle.so")

__all__ = ["Params", "Config", "EnvState", "StepOut", "lib", "build", "make_params", "make_config",
           "simple_solve", "llgs_solve", "resistance", "thermal_strength", "env_step", "env_step_batch",
           "thermal_normals", "parse_action", "simple_dmdt", "llgs_rhs", "DEV_TYPES", "sot_torque", "vcma_keff",
           "ArrayConfig", "make_array_config", "array_coupling", "ArrayEnvState", "array_step", "array_observation",
           "device_field", "ou_update"]

DEV_TYPES = {"stt_mram": 0, "sot_mram": 1, "vcma_mram": 2}


class Params(C.Structure):
    _fields_ = [("damping", C.c_double), ("ms", C.c_double), ("ku", C.c_double), ("volume", C.c_double),
                ("polarization", C.c_double), ("easy_axis", C.c_double * 3), ("demag", C.c_double * 3),
                ("a_ex", C.c_double), ("area", C.c_double), ("r_p", C.c_double), ("r_ap", C.c_double),
                ("ref_m", C.c_double * 3), ("r_series", C.c_double), ("sot_tau_dl", C.c_double), ("sot_tau_fl", C.c_double),
                ("sot_sigma", C.c_double * 3), ("vcma_xi", C.c_double), ("vcma_td", C.c_double), ("vcma_vbd", C.c_double),
                ("shape_demag", C.c_double * 3), ("dev_type", C.c_int32), ("params_valid", C.c_int32)]


class ArrayConfig(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("action_mode", C.c_int32), ("include_coupling", C.c_int32),
                ("max_steps", C.c_int32), ("obs_mode", C.c_int32), ("max_current", C.c_double),
                ("max_duration", C.c_double), ("success_threshold", C.c_double), ("energy_penalty_weight", C.c_double),
                ("temperature", C.c_double)]


class Config(C.Structure):
    _fields_ = [("solver", C.c_int32), ("thermal", C.c_int32), ("temperature", C.c_double),
                ("gamma", C.c_double), ("max_step", C.c_double), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_steps", C.c_int32), ("max_current", C.c_double), ("max_duration", C.c_double),
                ("success_threshold", C.c_double), ("energy_penalty_weight", C.c_double),
                ("seed", C.c_uint64), ("max_attempts", C.c_int64), ("torque_model", C.c_int32),
                ("noise_model", C.c_int32), ("noise_corr_time", C.c_double)]


class EnvState(C.Structure):
    _fields_ = [("m", C.c_double * 3), ("target", C.c_double * 3), ("total_energy", C.c_double),
                ("step_count", C.c_int32), ("rng_step", C.c_uint32), ("last_action", C.c_double * 2)]


class StepOut(C.Structure):
    _fields_ = [("obs", C.c_float * 12), ("reward", C.c_double), ("terminated", C.c_uint8),
                ("truncated", C.c_uint8), ("status", C.c_uint8), ("energy", C.c_double), ("n_sub", C.c_int32)]


def build(force=False):
    """Compile the oracle (gcc).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("stg_oracle.c", "stg_oracle.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.stgo_simple_solve.restype = C.c_int
        L.stgo_simple_solve.argtypes = [dp, C.c_double, C.POINTER(Params), C.POINTER(Config), C.c_double,
                                        C.c_uint64, C.c_uint32, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), dp, C.c_int64]
        L.stgo_llgs_solve.restype = C.c_int64
        L.stgo_llgs_solve.argtypes = [dp, C.c_double, C.POINTER(Params), C.POINTER(Config), C.c_double,
                                      C.c_uint64, C.c_uint32, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                      dp, dp, dp, dp, C.c_int64]
        L.stgo_resistance.restype = C.c_double
        L.stgo_resistance.argtypes = [dp, C.POINTER(Params)]
        L.stgo_thermal_strength.restype = C.c_double
        L.stgo_thermal_strength.argtypes = [C.POINTER(Params), C.c_double, C.c_double, C.c_int]
        L.stgo_parse_action.restype = None
        L.stgo_parse_action.argtypes = [C.POINTER(C.c_float), C.POINTER(Config), dp, dp]
        L.stgo_env_step.restype = None
        L.stgo_env_step.argtypes = [C.POINTER(EnvState), C.POINTER(C.c_float), C.POINTER(Params),
                                    C.POINTER(Config), C.c_uint64, C.POINTER(StepOut)]
        L.stgo_env_step_batch.restype = None
        L.stgo_env_step_batch.argtypes = [C.c_int64, C.POINTER(EnvState), C.POINTER(C.c_float),
                                          C.POINTER(Params), C.POINTER(C.c_uint8), C.POINTER(Config),
                                          C.c_uint64, C.POINTER(StepOut), C.c_int]
        L.stgo_thermal_normals.restype = None
        L.stgo_thermal_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, dp]
        L.stgo_simple_dmdt.restype = None
        L.stgo_simple_dmdt.argtypes = [dp, C.POINTER(Params), C.c_double, C.c_double, dp, dp]
        L.stgo_llgs_rhs.restype = None
        L.stgo_llgs_rhs.argtypes = [dp, C.POINTER(Params), C.c_double, C.c_double, dp, dp]
        L.stgo_sot_torque.restype = None
        L.stgo_sot_torque.argtypes = [dp, C.c_double, C.POINTER(Params), dp, dp]
        L.stgo_vcma_keff.restype = C.c_double
        L.stgo_vcma_keff.argtypes = [C.c_double, C.POINTER(Params)]
        L.stgo_array_coupling.restype = None
        L.stgo_array_coupling.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, dp]
        L.stgo_array_step.restype = None
        L.stgo_array_step.argtypes = [C.POINTER(ArrayConfig), C.POINTER(Params), dp, dp, dp, dp, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), dp, C.POINTER(C.c_uint8),
                                      C.POINTER(C.c_uint8), dp]
        L.stgo_array_observation.restype = None
        L.stgo_array_observation.argtypes = [C.POINTER(ArrayConfig), dp, dp, C.c_double, C.c_int32, C.POINTER(C.c_float)]
        L.stgo_device_field.restype = None
        L.stgo_device_field.argtypes = [dp, C.POINTER(Params), dp]
        L.stgo_ou_update.restype = None
        L.stgo_ou_update.argtypes = [dp, dp, C.c_double, C.c_double]
        L.stgo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _validate_as_stt(d):
    """utils/validation.py:176-234 validate_device_params(params, 'stt_mram') as a predicate."""
    def pos(v, lo):
        try:
            v = float(v)
        except (TypeError, ValueError):
            return False
        return np.isfinite(v) and v > 0 and v >= lo

    def prob(v):
        try:
            v = float(v)
        except (TypeError, ValueError):
            return False
        return np.isfinite(v) and 0 <= v <= 1
    for k in ("volume", "saturation_magnetization", "damping", "uniaxial_anisotropy", "easy_axis", "polarization"):
        if k not in d:
            return False
    e = np.asarray(d["easy_axis"], dtype=float)
    if e.shape != (3,) or not np.all(np.isfinite(e)) or np.linalg.norm(e) < 1e-12:
        return False
    return (pos(d["volume"], 1e-30) and pos(d["saturation_magnetization"], 1e3) and prob(d["damping"])
            and pos(d["uniaxial_anisotropy"], 1e3) and prob(d["polarization"]))


def make_params(device_params, device_type="stt_mram"):
    """Flatten a reference-style device_params dict.  Defaults are those the reference's .get()
    calls use (simple_solver.py:126-131, llgs_solver.py:79-82,192-205, spin_torque_env.py:476,502,
    devices/*.py)."""
    d = device_params
    p = Params()
    p.damping = d.get("damping", 0.01)
    p.ms = d.get("saturation_magnetization", 800e3)
    p.ku = d.get("uniaxial_anisotropy", 1e6)
    p.volume = d.get("volume", 1e-24)
    p.polarization = d.get("polarization", 0.7)
    p.easy_axis[:] = list(np.asarray(d.get("easy_axis", [0, 0, 1]), dtype=float))
    p.demag[:] = list(np.asarray(d.get("demag_factors", [0, 0, 1]), dtype=float))
    p.a_ex = d.get("exchange_constant", 20e-12)
    p.area = d.get("area", 1e-14)
    p.r_p = d.get("resistance_parallel", 1e3)
    p.r_ap = d.get("resistance_antiparallel", 2e3)
    p.ref_m[:] = list(np.asarray(d.get("reference_magnetization", [0, 0, 1]), dtype=float))
    p.dev_type = DEV_TYPES[device_type]
    p.r_series = 0.0
    # devices/sot_mram.py:114-132 (same in vcma_mram.py:149-166): demag factors from the aspect ratio
    p.shape_demag[:] = [0.0, 0.0, 0.0]
    if device_type in ("sot_mram", "vcma_mram"):
        ar = d.get("aspect_ratio", 1.0)
        if ar >= 1.0:
            n_x, n_y = 1.0 / (1.0 + ar), ar / (1.0 + ar)
        else:
            n_x, n_y = ar / (1.0 + ar), 1.0 / (1.0 + ar)
        p.shape_demag[:] = [n_x, n_y, 1.0 - n_x - n_y]
    p.sot_tau_dl = p.sot_tau_fl = 0.0
    p.sot_sigma[:] = [0.0, 1.0, 0.0]
    p.vcma_xi, p.vcma_td, p.vcma_vbd = 0.0, 1e-9, 2.0
    if device_type == "vcma_mram":
        # devices/vcma_mram.py:37-40
        p.vcma_xi = d.get("vcma_coefficient", 100e-6)
        p.vcma_td = d.get("dielectric_thickness", 1e-9)
        p.vcma_vbd = d.get("breakdown_voltage", 2.0)
    if device_type == "sot_mram":
        # devices/sot_mram.py:35-40,61-72,180-186
        sha = d.get("spin_hall_angle", 0.1)
        t_hm_ = d.get("heavy_metal_thickness", 5e-9)
        eff = sha * d.get("interface_transparency", 0.5) * (t_hm_ / (t_hm_ + d.get("thickness", 1e-9)))
        p.sot_tau_dl = d.get("damping_like_efficiency", 0.2) * eff
        p.sot_tau_fl = d.get("field_like_efficiency", 0.1) * eff
        j_hat = np.asarray(d.get("current_direction", [1.0, 0.0, 0.0]), dtype=float)
        j_hat = j_hat / np.linalg.norm(j_hat)
        p.sot_sigma[:] = list(np.cross(np.array([0.0, 0.0, 1.0]), j_hat))
    if device_type == "sot_mram":
        # devices/sot_mram.py:37-38,76-77,218-223
        t_hm = d.get("heavy_metal_thickness", 5e-9)
        rho = d.get("heavy_metal_resistivity", 2e-7)
        thickness = d.get("thickness", 1e-9)
        area = d.get("area", p.volume / thickness)
        p.r_series = (rho / t_hm) / (area * 1e-12) * 0.1
    p.params_valid = int(_validate_as_stt(d))
    return p


def make_config(solver="rk4", thermal=False, temperature=300.0, gamma=2.21e5, max_step=1e-12, rtol=1e-6,
                atol=1e-9, max_steps=100, max_current=2e6, max_duration=5e-9, success_threshold=0.9,
                energy_penalty_weight=0.1, seed=0, max_attempts=10_000_000, torque_model=0, noise_model=0,
                noise_corr_time=1e-12):
    c = Config()
    c.solver = {"rk4": 0, "euler": 1, "rk45": 2}[solver]
    c.thermal = int(bool(thermal))
    c.temperature, c.gamma, c.max_step, c.rtol, c.atol = temperature, gamma, max_step, rtol, atol
    c.max_steps, c.max_current, c.max_duration = max_steps, max_current, max_duration
    c.success_threshold, c.energy_penalty_weight = success_threshold, energy_penalty_weight
    c.seed, c.max_attempts = seed, max_attempts
    c.torque_model = int(torque_model)
    c.noise_model, c.noise_corr_time = int(noise_model), float(noise_corr_time)
    return c


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def simple_solve(m0, T, p, c, J, env_id=0, env_step=0, want_traj=False):
    m0 = np.ascontiguousarray(m0, dtype=np.float64)
    mf = np.zeros(3)
    n = C.c_int32(0)
    zr = C.c_int32(-1)
    nr = C.c_int32(0)
    traj = None
    cap = 0
    if want_traj:
        cap = 5002
        traj = np.zeros((cap, 3))
    ok = lib().stgo_simple_solve(_dp(m0), float(T), C.byref(p), C.byref(c), float(J), env_id, env_step,
                                 _dp(mf), C.byref(n), C.byref(zr), C.byref(nr),
                                 _dp(traj) if traj is not None else None, cap)
    out = dict(success=bool(ok), m_final=mf, n_steps=n.value, first_zero_row=zr.value, n_reset=nr.value)
    if want_traj:
        out["m"] = traj[: n.value + 1].copy()
    return out


def llgs_solve(m0, T, p, c, J, env_id=0, env_step=0, cap=20000):
    m0 = np.ascontiguousarray(m0, dtype=np.float64)
    mf = np.zeros(3)
    succ = C.c_int32(0)
    att = C.c_int64(0)
    t = np.zeros(cap)
    m = np.zeros((cap, 3))
    e = np.zeros(cap)
    tq = np.zeros(cap)
    n = lib().stgo_llgs_solve(_dp(m0), float(T), C.byref(p), C.byref(c), float(J), env_id, env_step, _dp(mf),
                              C.byref(succ), C.byref(att), _dp(t), _dp(m), _dp(e), _dp(tq), cap)
    k = min(n, cap)
    return dict(success=bool(succ.value), m_final=mf, n_points=int(n), n_attempts=att.value,
                t=t[:k].copy(), m=m[:k].copy(), energy=e[:k].copy(), torques=tq[:k].copy())


def resistance(m, p):
    m = np.ascontiguousarray(m, dtype=np.float64)
    return lib().stgo_resistance(_dp(m), C.byref(p))


def thermal_strength(p, gamma, temperature, which):
    return lib().stgo_thermal_strength(C.byref(p), gamma, temperature, which)


def thermal_normals(seed, env_id, env_step, call_idx):
    z = np.zeros(3)
    lib().stgo_thermal_normals(seed, env_id, env_step, call_idx, _dp(z))
    return z


def parse_action(action, c):
    a = np.ascontiguousarray(action, dtype=np.float32)
    J = C.c_double()
    T = C.c_double()
    lib().stgo_parse_action(a.ctypes.data_as(C.POINTER(C.c_float)), C.byref(c), C.byref(J), C.byref(T))
    return J.value, T.value


def simple_dmdt(m, p, gamma, J, h_thermal=None):
    m = np.ascontiguousarray(m, dtype=np.float64)
    out = np.zeros(3)
    h = None if h_thermal is None else np.ascontiguousarray(h_thermal, dtype=np.float64)
    lib().stgo_simple_dmdt(_dp(m), C.byref(p), gamma, J, _dp(h) if h is not None else None, _dp(out))
    return out


def llgs_rhs(y, p, gamma, J, h_thermal=None):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(3)
    h = None if h_thermal is None else np.ascontiguousarray(h_thermal, dtype=np.float64)
    lib().stgo_llgs_rhs(_dp(y), C.byref(p), gamma, J, _dp(h) if h is not None else None, _dp(out))
    return out


def sot_torque(m, J, p):
    m = np.ascontiguousarray(m, dtype=np.float64)
    dl, fl = np.zeros(3), np.zeros(3)
    lib().stgo_sot_torque(_dp(m), float(J), C.byref(p), _dp(dl), _dp(fl))
    return dl, fl


def vcma_keff(volt, p):
    return lib().stgo_vcma_keff(float(volt), C.byref(p))


ARRAY_MODES = {"individual": 0, "row": 1, "column": 2, "global": 3}
COUPLING_TYPES = {"dipolar": 0, "exchange": 1, "stray_field": 2}


def make_array_config(rows=4, cols=4, action_mode="individual", include_coupling=True, max_steps=200, obs_mode="array",
                      max_current=2e6, max_duration=5e-9, success_threshold=0.9, energy_penalty_weight=0.1,
                      temperature=300.0):
    c = ArrayConfig()
    c.rows, c.cols, c.action_mode = rows, cols, ARRAY_MODES[action_mode]
    c.include_coupling, c.max_steps, c.obs_mode = int(include_coupling), max_steps, {"array": 0, "vector": 1}[obs_mode]
    c.max_current, c.max_duration, c.success_threshold = max_current, max_duration, success_threshold
    c.energy_penalty_weight, c.temperature = energy_penalty_weight, temperature
    return c


def array_coupling(rows, cols, coupling_type="dipolar", strength=0.1):
    out = np.zeros((rows * cols, rows * cols))
    lib().stgo_array_coupling(rows, cols, COUPLING_TYPES[coupling_type], float(strength), _dp(out))
    return out


class ArrayEnvState:
    """Mutable state of one SpinTorqueArray-v0 env for the oracle."""

    def __init__(self, pattern, target):
        self.pattern = np.ascontiguousarray(np.asarray(pattern, dtype=np.float64).reshape(-1, 3))
        self.target = np.ascontiguousarray(np.asarray(target, dtype=np.float64).reshape(-1, 3))
        self.total_energy = C.c_double(0.0)
        self.step_count = C.c_int32(0)


def array_step(state, action, p, c, coupling):
    a = np.ascontiguousarray(action, dtype=np.float32)
    n = c.rows * c.cols
    obs = np.zeros(n * 6 + (4 if c.obs_mode == 1 else 0), dtype=np.float32)
    rew, en = C.c_double(), C.c_double()
    te, tr = C.c_uint8(), C.c_uint8()
    coupling = np.ascontiguousarray(coupling, dtype=np.float64)
    lib().stgo_array_step(C.byref(c), C.byref(p), _dp(coupling), _dp(state.pattern), _dp(state.target),
                          C.byref(state.total_energy), C.byref(state.step_count), a.ctypes.data_as(C.POINTER(C.c_float)),
                          len(a), obs.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rew), C.byref(te), C.byref(tr),
                          C.byref(en))
    return obs, rew.value, bool(te.value), bool(tr.value), en.value


def array_observation(state, c):
    n = c.rows * c.cols
    obs = np.zeros(n * 6 + (4 if c.obs_mode == 1 else 0), dtype=np.float32)
    lib().stgo_array_observation(C.byref(c), _dp(state.pattern), _dp(state.target), state.total_energy.value,
                                 state.step_count.value, obs.ctypes.data_as(C.POINTER(C.c_float)))
    return obs


def device_field(m, p):
    m = np.ascontiguousarray(m, dtype=np.float64)
    h = np.zeros(3)
    lib().stgo_device_field(_dp(m), C.byref(p), _dp(h))
    return h


def env_step(state, action, p, c, env_id=0):
    a = np.ascontiguousarray(action, dtype=np.float32)
    out = StepOut()
    lib().stgo_env_step(C.byref(state), a.ctypes.data_as(C.POINTER(C.c_float)), C.byref(p), C.byref(c),
                        env_id, C.byref(out))
    return out


def env_step_batch(states, actions, params, cls, c, env_id0=0, n_threads=0):
    """states: (EnvState * n) ctypes array; actions: float32 [n,2]; params: (Params * k) array;
    cls: uint8 [n] or None.  Returns a (StepOut * n) array."""
    n = len(states)
    a = np.ascontiguousarray(actions, dtype=np.float32)
    assert a.shape == (n, 2)
    outs = (StepOut * n)()
    cp = None
    if cls is not None:
        cls = np.ascontiguousarray(cls, dtype=np.uint8)
        cp = cls.ctypes.data_as(C.POINTER(C.c_uint8))
    lib().stgo_env_step_batch(n, states, a.ctypes.data_as(C.POINTER(C.c_float)), params, cp, C.byref(c),
                              env_id0, outs, n_threads)
    return outs


def ou_update(x, xi, dt, corr_time):
    """ThermalFluctuations._generate_correlated_noise state update (thermal_model.py:113-137); returns the new x."""
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    lib().stgo_ou_update(_dp(x), _dp(xi), float(dt), float(corr_time))
    return x
