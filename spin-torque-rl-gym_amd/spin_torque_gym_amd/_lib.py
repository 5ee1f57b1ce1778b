"""ctypes binding of libspintorque_hip.so (include/spintorque_hip.h).

The library is the product: there is no CPU fallback.  Loading fails loudly (ImportError/RuntimeError)
when the shared object has not been built (run ``python __graft_entry__.py`` or
``make -C spin-torque-rl-gym_amd/csrc``), and every compute entry point needs a visible MI355X.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STG_HIP_LIBRARY points at another build of the same library (A/B runs of kernel changes); the default is the in-tree one
LIB_PATH = os.environ.get("STG_HIP_LIBRARY") or os.path.join(_HERE, "libspintorque_hip.so")

STG_MAX_TARGETS = 8
STG_MAX_CLASSES = 64
ABI_VERSION = 4          # STG_ABI_VERSION of include/spintorque_hip.h this binding was written against
STG_NPARAM = 30          # double-valued fields of stg_device_params, in declaration order
SOLVERS = {"rk4": 0, "euler": 1, "rk45": 2}
DEV_TYPES = {"stt_mram": 0, "sot_mram": 1, "vcma_mram": 2}
OUT_LAYOUTS = {"soa": 0, "records": 1}
RECORD_BYTES = 56        # STG_RECORD_BYTES
STATUS_OK, STATUS_NOOP, STATUS_RESET, STATUS_INACTIVE = 0, 1, 2, 3
STG_OK, STG_E_INVALID, STG_E_HIP, STG_E_NOMEM, STG_E_STATE = 0, -1, -2, -3, -4


class StgConfig(C.Structure):
    _fields_ = [
        ("solver", C.c_int32), ("thermal", C.c_int32), ("temperature", C.c_double), ("gamma", C.c_double),
        ("max_step", C.c_double), ("rtol", C.c_double), ("atol", C.c_double), ("max_steps", C.c_int32),
        ("n_targets", C.c_int32), ("max_current", C.c_double), ("max_duration", C.c_double),
        ("success_threshold", C.c_double), ("energy_penalty_weight", C.c_double),
        ("targets", (C.c_double * 3) * STG_MAX_TARGETS), ("seed", C.c_uint64), ("max_attempts", C.c_int64),
        ("skip_done", C.c_int32), ("torque_model", C.c_int32), ("wave_spec", C.c_int32), ("lane_sort", C.c_int32),
        ("noise_model", C.c_int32), ("out_layout", C.c_int32), ("noise_corr_time", C.c_double),
        ("lane_refill", C.c_int32), ("reserved0", C.c_int32),
    ]


class StgDeviceParams(C.Structure):
    _fields_ = [
        ("damping", C.c_double), ("ms", C.c_double), ("ku", C.c_double), ("volume", C.c_double),
        ("polarization", C.c_double), ("easy_axis", C.c_double * 3), ("demag", C.c_double * 3),
        ("a_ex", C.c_double), ("area", C.c_double), ("r_p", C.c_double), ("r_ap", C.c_double),
        ("ref_m", C.c_double * 3), ("r_series", C.c_double), ("sot_tau_dl", C.c_double), ("sot_tau_fl", C.c_double),
        ("sot_sigma", C.c_double * 3), ("vcma_xi", C.c_double), ("vcma_td", C.c_double), ("vcma_vbd", C.c_double),
        ("shape_demag", C.c_double * 3), ("dev_type", C.c_int32), ("params_valid", C.c_int32),
    ]


class StgArrayConfig(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("action_mode", C.c_int32), ("include_coupling", C.c_int32),
                ("max_steps", C.c_int32), ("obs_mode", C.c_int32), ("max_current", C.c_double),
                ("max_duration", C.c_double), ("success_threshold", C.c_double), ("energy_penalty_weight", C.c_double),
                ("temperature", C.c_double)]


# every symbol include/spintorque_hip.h declares: (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "stg_create": (C.c_int, [C.POINTER(_VP), C.c_int, C.c_int64, C.c_int64, C.POINTER(StgConfig)]),
    "stg_destroy": (None, [_VP]),
    "stg_last_error": (C.c_char_p, []),
    "stg_abi_version": (C.c_int, []),
    "stg_set_params": (C.c_int, [_VP, C.POINTER(StgDeviceParams), C.c_int32, _VP]),
    "stg_set_params_per_env": (C.c_int, [_VP, _VP, _VP, _VP]),
    "stg_reset": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint64, _VP, _VP]),
    "stg_step": (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_step_many": (C.c_int, [_VP, C.c_int32, _VP, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_get_state": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_set_state": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_device_terms": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_get_counters": (C.c_int, [_VP, C.POINTER(C.c_uint64), C.c_int32]),
    "stg_get_placement": (C.c_int, [_VP, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stg_solve": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint32, _VP, _VP, _VP, _VP]),
    "stg_solve_traj": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint32, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_thermal_strength": (C.c_int, [_VP, C.c_int32, C.POINTER(C.c_double)]),
    "stg_thermal_normals": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_int32, _VP, _VP]),
    "stg_array_create": (C.c_int, [C.POINTER(_VP), C.c_int, C.c_int64, C.c_int64, C.POINTER(StgArrayConfig),
                                   C.POINTER(StgDeviceParams), C.POINTER(C.c_double)]),
    "stg_array_destroy": (None, [_VP]),
    "stg_array_reset": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint64, _VP, _VP]),
    "stg_array_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "stg_array_get_state": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built.  Build it with "
            f"`python __graft_entry__.py` (or `make -C spin-torque-rl-gym_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.stg_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI version mismatch: library reports {lib.stg_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


class StgError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().stg_last_error()
        raise StgError(f"libspintorque_hip error {rc}: {msg.decode() if msg else '?'}")
