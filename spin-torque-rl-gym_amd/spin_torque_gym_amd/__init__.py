"""spin_torque_gym_amd -- MI355X-native SpinTorque-v0 step path.

Host-side mirror of the parts of ``spin_torque_gym`` that sit on ``SpinTorqueEnv.step()``:

    from spin_torque_gym_amd import SpinTorqueEnv, SpinTorqueVecEnv, DeviceFactory
    from spin_torque_gym_amd.physics import LLGSSolver, SimpleLLGSSolver, RobustLLGSSolver, ThermalFluctuations

All physics runs in ``libspintorque_hip.so`` (hand-written HIP for gfx950, include/spintorque_hip.h); importing
this package does not need a GPU, constructing an environment or solver does.
"""
from .backend import EnvConfig, HipBackend
from .devices import (BaseSpintronicDevice, DeviceFactory, SOTMRAMDevice, STTMRAMDevice, VCMAMRAMDevice,
                      flatten_params, params_valid_as_stt)
from .array_env import SpinTorqueArrayEnv, SpinTorqueArrayVecEnv
from .envs import SpinTorqueEnv, SpinTorqueVecEnv, register_envs

__version__ = "0.1.0"
__all__ = ["EnvConfig", "HipBackend", "BaseSpintronicDevice", "DeviceFactory", "STTMRAMDevice", "SOTMRAMDevice",
           "VCMAMRAMDevice", "flatten_params", "params_valid_as_stt", "SpinTorqueEnv", "SpinTorqueVecEnv",
           "register_envs", "SpinTorqueArrayEnv", "SpinTorqueArrayVecEnv"]

register_envs()
