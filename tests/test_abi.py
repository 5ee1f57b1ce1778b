"""The C-ABI library loads and exports every symbol include/spintorque_hip.h declares (no compute: CPU only)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spintorque_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stg_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from spin_torque_gym_amd import _lib
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert set(declared) == set(_lib.SYMBOLS), (set(declared) ^ set(_lib.SYMBOLS))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert lib.stg_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_the_header():
    """sizeof of the ctypes mirrors equals what the C compiler computes for the header's structs."""
    import subprocess
    import tempfile
    from spin_torque_gym_amd import _lib
    src = '#include <stdio.h>\n#include "spintorque_hip.h"\nint main(){printf("%zu %zu\\n", sizeof(stg_config), sizeof(stg_device_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert C.sizeof(_lib.StgConfig) == a
    assert C.sizeof(_lib.StgDeviceParams) == b


def test_argument_errors_without_gpu():
    """Entry points validate their arguments before touching the device."""
    from spin_torque_gym_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.stg_create(C.byref(ctx), 0, 0, 0, None) == -1          # n_envs < 1
    assert b"n_envs" in lib.stg_last_error()
    cfg = _lib.StgConfig()
    cfg.solver = 7
    assert lib.stg_create(C.byref(ctx), 0, 16, 0, C.byref(cfg)) == -1
    assert b"solver" in lib.stg_last_error()
    assert lib.stg_step(None, None, 0, None, None, None, None, None, None, None, None) == -1


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import spin_torque_gym_amd as s
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.SpinTorqueVecEnv(8)
