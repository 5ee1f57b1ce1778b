"""Device parameter surface: the `spin_torque_gym.devices` API the env path consumes.

Mirrors (same names, argument meaning and error behaviour):
  DeviceFactory ............ spin_torque_gym/devices/device_factory.py:18-194
  BaseSpintronicDevice ..... spin_torque_gym/devices/base_device.py:13-138
  STTMRAMDevice ............ spin_torque_gym/devices/stt_mram.py:16-98
  SOTMRAMDevice ............ spin_torque_gym/devices/sot_mram.py:16-228 (parameters, resistance, torque factors)
  VCMAMRAMDevice ........... spin_torque_gym/devices/vcma_mram.py:16-257 (parameters, resistance, K_eff)

The env only reads ``device.device_params``, ``get_parameter`` and ``compute_resistance``
(spin_torque_env.py:454,475,501); on the GPU path those values travel as one flattened
``stg_device_params`` record per device class (`flatten_params`) and the resistance is evaluated in the
step kernel.  The host-side ``compute_resistance`` here is API surface for user code (analysis, tests), it is
never called from step()/reset().  Skyrmion devices are out of scope (different physics).
"""
import warnings
from typing import Any, Dict

import numpy as np

from . import _lib

MU0 = 4 * np.pi * 1e-7


class BaseSpintronicDevice:
    """Common parameter handling (base_device.py:13-138)."""

    device_type = "base"
    _required = ("volume", "saturation_magnetization")

    def __init__(self, device_params: Dict[str, Any]):
        self.device_params = device_params.copy()
        self.volume = device_params.get("volume", 1e-24)
        self.thickness = device_params.get("thickness", 1e-9)
        self.saturation_magnetization = device_params.get("saturation_magnetization", 800e3)
        self.mu0 = MU0
        self.kb = 1.380649e-23
        self.e = 1.602176634e-19
        self.hbar = 1.054571817e-34
        self._validate_parameters()

    def _validate_parameters(self) -> None:
        for key in self._required:
            if key not in self.device_params:
                raise ValueError(f"Missing required parameter: {key}")

    def get_parameter(self, key: str, default: Any = None) -> Any:
        return self.device_params.get(key, default)

    def set_parameter(self, key: str, value: Any) -> None:
        self.device_params[key] = value

    def validate_magnetization(self, magnetization) -> np.ndarray:
        """base_device.py:94-116: unit vector or ValueError."""
        if not isinstance(magnetization, np.ndarray):
            magnetization = np.array(magnetization)
        if magnetization.shape != (3,):
            raise ValueError(f"Magnetization must be 3D vector, got shape {magnetization.shape}")
        magnitude = np.linalg.norm(magnetization)
        if magnitude < 1e-12:
            raise ValueError("Magnetization vector cannot be zero")
        return magnetization / magnitude

    def _reference_layer(self) -> np.ndarray:
        ref = np.asarray(self.device_params.get("reference_magnetization", np.array([0, 0, 1])), dtype=float)
        return ref / np.linalg.norm(ref)

    def compute_resistance(self, magnetization: np.ndarray) -> float:
        raise NotImplementedError

    def get_device_info(self) -> Dict[str, Any]:
        return {
            "device_type": self.__class__.__name__,
            "volume": self.volume,
            "thickness": self.thickness,
            "saturation_magnetization": self.saturation_magnetization,
            "parameters": self.device_params.copy(),
        }

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(volume={self.volume:.2e}, Ms={self.saturation_magnetization:.0f})"


class STTMRAMDevice(BaseSpintronicDevice):
    """stt_mram.py:16-98."""

    device_type = "stt_mram"
    _required = ("volume", "saturation_magnetization", "damping", "uniaxial_anisotropy", "polarization")

    def __init__(self, device_params):
        super().__init__(device_params)
        self.reference_magnetization = self.validate_magnetization(
            self.get_parameter("reference_magnetization", np.array([0, 0, 1])))

    def _validate_parameters(self) -> None:
        super()._validate_parameters()
        p = self.device_params
        if p["volume"] <= 0:
            raise ValueError("Volume must be positive")
        if p["saturation_magnetization"] <= 0:
            raise ValueError("Saturation magnetization must be positive")
        if not 0 <= p["damping"] <= 1:
            raise ValueError("Damping must be between 0 and 1")
        if not 0 <= p["polarization"] <= 1:
            raise ValueError("Polarization must be between 0 and 1")

    def compute_effective_field(self, magnetization, applied_field):
        """stt_mram.py:55-76: applied field + uniaxial anisotropy of the re-normalised magnetisation."""
        m = self.validate_magnetization(magnetization)
        e = self.get_parameter("easy_axis", np.array([0, 0, 1]))
        k_u = self.get_parameter("uniaxial_anisotropy", 1e6)
        ms = self.get_parameter("saturation_magnetization", 800e3)
        return np.asarray(applied_field, dtype=float) + (2 * k_u / (self.mu0 * ms)) * np.dot(m, e) * e

    def compute_resistance(self, magnetization) -> float:
        """stt_mram.py:78-94: R_P (1 + TMR (1 - cos)/2), floor R_P/2; the input is re-normalised."""
        m = self.validate_magnetization(magnetization)
        r_p = self.get_parameter("resistance_parallel", 1e3)
        r_ap = self.get_parameter("resistance_antiparallel", 2e3)
        tmr = (r_ap - r_p) / r_p
        r = r_p * (1 + tmr * (1 - np.dot(m, self.reference_magnetization)) / 2)
        return max(r, r_p * 0.5)


class SOTMRAMDevice(BaseSpintronicDevice):
    """sot_mram.py:16-228 (parameter surface, torque efficiency factors, resistance)."""

    device_type = "sot_mram"
    _required = ("volume", "saturation_magnetization", "damping", "uniaxial_anisotropy", "easy_axis")

    def __init__(self, device_params):
        super().__init__(device_params)
        p = device_params
        if p.get("spin_hall_angle", 0.1) > 1.0:
            warnings.warn("Spin Hall angle > 1.0 is physically unrealistic")
        self.spin_hall_angle = p.get("spin_hall_angle", 0.1)
        self.heavy_metal_thickness = p.get("heavy_metal_thickness", 5e-9)
        self.heavy_metal_resistivity = p.get("heavy_metal_resistivity", 2e-7)
        self.interface_transparency = p.get("interface_transparency", 0.5)
        self.field_like_efficiency = p.get("field_like_efficiency", 0.1)
        self.damping_like_efficiency = p.get("damping_like_efficiency", 0.2)
        # sot_mram.py:61-77
        self.j_s_efficiency = (self.spin_hall_angle * self.interface_transparency
                               * (self.heavy_metal_thickness / (self.heavy_metal_thickness + self.thickness)))
        self.tau_dl_factor = self.damping_like_efficiency * self.j_s_efficiency
        self.tau_fl_factor = self.field_like_efficiency * self.j_s_efficiency
        self.sheet_resistance_hm = self.heavy_metal_resistivity / self.heavy_metal_thickness
        self.area = p.get("area", self.volume / self.thickness)

    def series_resistance(self) -> float:
        """0.1 * r_hm with r_hm = sheet_resistance_hm / (area * 1e-12) (sot_mram.py:218-223)."""
        return (self.sheet_resistance_hm / (self.area * 1e-12)) * 0.1

    def compute_resistance(self, magnetization) -> float:
        """sot_mram.py:196-228: R_P + (R_AP - R_P)(1 - cos)/2 + 0.1 r_hm, floor 1; no re-normalisation."""
        r_p = self.device_params.get("resistance_parallel", 1e3)
        r_ap = self.device_params.get("resistance_antiparallel", 2e3)
        r_mtj = r_p + (r_ap - r_p) * (1 - np.dot(magnetization, self._reference_layer())) / 2
        return max(r_mtj + self.series_resistance(), 1.0)

    def demag_factors(self):
        return _shape_demag(self.device_params.get("aspect_ratio", 1.0))

    def compute_effective_field(self, magnetization, applied_field):
        """sot_mram.py:79-112 with zero temperature/DMI: applied + anisotropy + shape demagnetisation."""
        return _thin_film_field(self, magnetization, applied_field)

    def compute_spin_torque(self, current_density, magnetization, current_direction=None):
        """sot_mram.py:163-194: tau_DL = f_dl J (sigma x m), tau_FL = f_fl J sigma, sigma = z x j_hat."""
        j_hat = np.array([1.0, 0.0, 0.0]) if current_direction is None else np.asarray(current_direction, dtype=float)
        j_hat = j_hat / np.linalg.norm(j_hat)
        sigma = np.cross(np.array([0.0, 0.0, 1.0]), j_hat)
        return (self.tau_dl_factor * current_density * np.cross(sigma, magnetization),
                self.tau_fl_factor * current_density * sigma)

    # -- analysis helpers of the device class (host closed forms, sot_mram.py:230-281,363-435) ----------------------------
    def compute_power_consumption(self, current_density: float, pulse_duration: float, magnetization=None) -> float:
        """sot_mram.py:230-255: Joule energy in the heavy-metal line, I^2 R_hm t with R_hm = sheet/(area 1e-12)."""
        if abs(current_density) < 1e-12:
            return 0.0
        current = current_density * self.area
        voltage = current * self.sheet_resistance_hm / (self.area * 1e-12)
        return voltage * current * pulse_duration

    def get_switching_threshold(self) -> Dict[str, float]:
        """sot_mram.py:257-281: the class's empirical critical current density and the anisotropy field."""
        alpha = self.device_params["damping"]
        h_k = 2 * self.device_params["uniaxial_anisotropy"] / (self.mu0 * self.saturation_magnetization)
        j_c = 5e6 * (1 + alpha) * (1 + h_k / 1e6) / (1 + self.tau_dl_factor)
        return {"critical_current_density": j_c, "critical_field": h_k,
                "damping_like_efficiency": self.tau_dl_factor, "field_like_efficiency": self.tau_fl_factor}

    def update_temperature(self, temperature: float) -> None:
        """sot_mram.py:363-374."""
        self.device_params["temperature"] = temperature
        if temperature > 400:
            warnings.warn(f"High temperature ({temperature} K) may affect device reliability")

    def compute_energy_barrier(self, magnetization) -> float:
        """sot_mram.py:376-395: K V (1 - cos^2) against the (raw) easy axis."""
        c = abs(np.dot(magnetization, self.device_params["easy_axis"]))
        return self.device_params["uniaxial_anisotropy"] * self.volume * (1 - c ** 2)

    def estimate_switching_time(self, current_density: float, temperature: float = 300.0) -> float:
        """sot_mram.py:397-435: Arrhenius below the critical current (current-assisted barrier), pi alpha / (gamma f_dl J)
        above it."""
        if abs(current_density) < 1e-6:
            return np.inf
        j_c = self.get_switching_threshold()["critical_current_density"]
        if abs(current_density) < j_c:
            e_b = self.device_params["uniaxial_anisotropy"] * self.volume
            return (1 / 1e9) * np.exp(e_b / (1.38e-23 * temperature) * (1 - abs(current_density) / j_c))
        return (np.pi * self.device_params["damping"]) / (2.21e5 * self.tau_dl_factor * abs(current_density))


class VCMAMRAMDevice(BaseSpintronicDevice):
    """vcma_mram.py:16-257 (parameter surface, effective anisotropy, resistance)."""

    device_type = "vcma_mram"
    _required = ("volume", "saturation_magnetization", "damping", "uniaxial_anisotropy", "easy_axis")

    def __init__(self, device_params):
        super().__init__(device_params)
        p = device_params
        if p.get("vcma_coefficient", 100e-6) < 0:
            warnings.warn("Negative VCMA coefficient indicates inverted VCMA effect")
        self.vcma_coefficient = p.get("vcma_coefficient", 100e-6)
        self.dielectric_thickness = p.get("dielectric_thickness", 1e-9)
        self.dielectric_constant = p.get("dielectric_constant", 25.0)
        self.breakdown_voltage = p.get("breakdown_voltage", 2.0)
        self.leakage_resistance = p.get("leakage_resistance", 1e12)
        self.area = p.get("area", self.volume / self.thickness)
        self.base_anisotropy = p["uniaxial_anisotropy"]

    def effective_anisotropy(self, voltage: float) -> float:
        """vcma_mram.py:122-147: K - xi |V| / t_d^2, V clipped to +-V_bd, floor -K/2."""
        v = float(np.clip(voltage, -self.breakdown_voltage, self.breakdown_voltage))
        k_eff = self.base_anisotropy - self.vcma_coefficient * abs(v) / (self.dielectric_thickness ** 2)
        return max(k_eff, -0.5 * self.base_anisotropy)

    _compute_effective_anisotropy = effective_anisotropy

    def demag_factors(self):
        return _shape_demag(self.device_params.get("aspect_ratio", 1.0))

    def compute_effective_field(self, magnetization, applied_field, applied_voltage: float = 0.0):
        """vcma_mram.py:85-120 with zero temperature: applied + K_eff(V) anisotropy + shape demagnetisation."""
        return _thin_film_field(self, magnetization, applied_field, self.effective_anisotropy(applied_voltage))

    def compute_resistance(self, magnetization) -> float:
        """vcma_mram.py:236-257: R_P + (R_AP - R_P)(1 - cos)/2, floor 1; no re-normalisation."""
        r_p = self.device_params.get("resistance_parallel", 1e3)
        r_ap = self.device_params.get("resistance_antiparallel", 2e3)
        r = r_p + (r_ap - r_p) * (1 - np.dot(magnetization, self._reference_layer())) / 2
        return max(r, 1.0)

    # -- analysis helpers of the device class (host closed forms, vcma_mram.py:60-84,187-234,259-320,400-504) -------------
    @property
    def capacitance(self) -> float:
        """vcma_mram.py:62-70: eps0 eps_r A / t_d unless 'capacitance_per_area' is given."""
        cpa = self.device_params.get("capacitance_per_area")
        return (8.854e-12 * self.dielectric_constant * self.area / self.dielectric_thickness) if cpa is None else cpa * self.area

    def compute_switching_probability(self, voltage: float, pulse_duration: float, temperature: float = 300.0,
                                      initial_state=None) -> float:
        """vcma_mram.py:187-234: 1 - exp(-f0 exp(-K_eff(V) V / k_B T) t); 1 when the barrier is gone."""
        e_b = self.effective_anisotropy(voltage) * self.volume
        kt = 1.38e-23 * temperature
        if e_b <= 0:
            return 1.0
        if kt <= 0:
            return 0.0
        return min(1.0 - np.exp(-(1e9 * np.exp(-e_b / kt)) * pulse_duration), 1.0)

    def compute_power_consumption(self, voltage: float, pulse_duration: float, magnetization=None) -> float:
        """vcma_mram.py:259-287: capacitive charging + ohmic leakage."""
        if abs(voltage) < 1e-12:
            return 0.0
        return 0.5 * self.capacitance * voltage ** 2 + voltage ** 2 * pulse_duration / self.leakage_resistance

    def get_switching_threshold(self) -> Dict[str, float]:
        """vcma_mram.py:289-320: voltage that cancels the anisotropy (as the class computes it, with the free-layer
        thickness) and the voltage that leaves a 40 k_B T barrier, both clipped to breakdown."""
        v_c = min(abs(self.base_anisotropy * self.thickness / self.vcma_coefficient), self.breakdown_voltage)
        kt = 1.38e-23 * self.device_params.get("temperature", 300.0)
        need = self.base_anisotropy * self.volume - 40 * kt
        v_t = (need / (self.vcma_coefficient * self.area)) * self.dielectric_thickness if need > 0 else 0.0
        return {"critical_voltage": v_c, "thermal_switching_voltage": min(v_t, self.breakdown_voltage),
                "breakdown_voltage": self.breakdown_voltage, "vcma_coefficient": self.vcma_coefficient}

    def update_temperature(self, temperature: float) -> None:
        """vcma_mram.py:400-413."""
        self.device_params["temperature"] = temperature
        self.thermal_energy = 1.38e-23 * temperature
        if temperature > 400:
            warnings.warn(f"High temperature ({temperature} K) may affect VCMA efficiency")

    def compute_energy_barrier(self, magnetization, voltage: float = 0.0) -> float:
        """vcma_mram.py:415-442: |K_eff(V)| V."""
        return max(abs(self.effective_anisotropy(voltage)) * self.volume, 0.0)

    def estimate_switching_time(self, voltage: float, temperature: float = 300.0) -> float:
        """vcma_mram.py:444-478."""
        if abs(voltage) < 1e-6:
            return np.inf
        e_b = self.effective_anisotropy(voltage) * self.volume
        if e_b <= 0:
            return 1e-12
        return (1 / 1e9) * np.exp(e_b / (1.38e-23 * temperature))

    def compute_leakage_current(self, voltage: float) -> float:
        """vcma_mram.py:480-504: ohmic leakage plus the class's simplified Fowler-Nordheim term above 1e8 V/m."""
        if abs(voltage) < 1e-12:
            return 0.0
        current = voltage / self.leakage_resistance
        field = abs(voltage) / self.dielectric_thickness
        if field > 1e8:
            current += 1e-6 * field * np.exp(-3.5e9 / field) * self.area
        return current


def _shape_demag(aspect_ratio: float):
    """sot_mram.py:114-132 / vcma_mram.py:149-166: demagnetisation factors of an elliptical thin film."""
    if aspect_ratio >= 1.0:
        n_x, n_y = 1.0 / (1.0 + aspect_ratio), aspect_ratio / (1.0 + aspect_ratio)
    else:
        n_x, n_y = aspect_ratio / (1.0 + aspect_ratio), 1.0 / (1.0 + aspect_ratio)
    return np.array([n_x, n_y, 1.0 - n_x - n_y])


def _thin_film_field(device, magnetization, applied_field, k_u=None):
    e = device.device_params["easy_axis"]
    k_u = device.device_params["uniaxial_anisotropy"] if k_u is None else k_u
    ms = device.saturation_magnetization
    h_anis = (2 * k_u / (device.mu0 * ms)) * np.dot(magnetization, e) * e
    return applied_field + h_anis + (-ms * device.demag_factors() * magnetization)


_DEFAULTS = {
    # device_factory.py:129-172 (values are data)
    "stt_mram": lambda: {
        "volume": 50e-9 * 100e-9 * 2e-9, "area": 50e-9 * 100e-9, "thickness": 2e-9, "aspect_ratio": 2.0,
        "saturation_magnetization": 800e3, "damping": 0.01, "uniaxial_anisotropy": 1.2e6,
        "exchange_constant": 20e-12, "polarization": 0.7, "resistance_parallel": 1e3,
        "resistance_antiparallel": 2e3, "easy_axis": np.array([0, 0, 1]),
        "reference_magnetization": np.array([0, 0, 1])},
    "sot_mram": lambda: {
        "volume": 100e-9 * 100e-9 * 1e-9, "area": 100e-9 * 100e-9, "thickness": 1e-9,
        "saturation_magnetization": 800e3, "damping": 0.015, "uniaxial_anisotropy": 0.8e6,
        "exchange_constant": 20e-12, "spin_hall_angle": 0.2, "resistance_parallel": 500,
        "resistance_antiparallel": 1000, "easy_axis": np.array([0, 0, 1])},
    "vcma_mram": lambda: {
        "volume": 80e-9 * 80e-9 * 1.5e-9, "area": 80e-9 * 80e-9, "thickness": 1.5e-9,
        "saturation_magnetization": 800e3, "damping": 0.008, "uniaxial_anisotropy": 1.5e6,
        "exchange_constant": 20e-12, "vcma_coefficient": 100e-6, "resistance_parallel": 2e3,
        "resistance_antiparallel": 4e3, "easy_axis": np.array([0, 0, 1])},
}
_GENERIC = lambda: {"volume": 1e-24, "saturation_magnetization": 800e3, "damping": 0.01,  # noqa: E731
                    "uniaxial_anisotropy": 1e6, "exchange_constant": 20e-12, "polarization": 0.7}


class DeviceFactory:
    """device_factory.py:18-194."""

    def __init__(self):
        self._device_types = {}
        for name, cls in (("stt_mram", STTMRAMDevice), ("sot_mram", SOTMRAMDevice), ("vcma_mram", VCMAMRAMDevice)):
            self.register_device(name, cls)

    def register_device(self, device_type: str, device_class) -> None:
        if not (isinstance(device_class, type) and issubclass(device_class, BaseSpintronicDevice)):
            raise ValueError("Device class must inherit from BaseSpintronicDevice")
        self._device_types[device_type.lower()] = device_class

    def create_device(self, device_type: str, device_params: Dict[str, Any]) -> BaseSpintronicDevice:
        device_type = device_type.lower()
        if device_type not in self._device_types:
            raise ValueError(f"Unknown device type '{device_type}'. Available types: {list(self._device_types)}")
        try:
            return self._device_types[device_type](device_params)
        except Exception as e:      # device_factory.py:74-77
            raise RuntimeError(f"Failed to create {device_type} device: {e}")

    def get_available_devices(self) -> list:
        return list(self._device_types)

    def get_device_info(self, device_type: str) -> Dict[str, Any]:
        device_type = device_type.lower()
        if device_type not in self._device_types:
            raise ValueError(f"Unknown device type '{device_type}'")
        cls = self._device_types[device_type]
        return {"name": device_type, "class": cls.__name__, "module": cls.__module__, "docstring": cls.__doc__}

    def get_default_parameters(self, device_type: str) -> Dict[str, Any]:
        return _DEFAULTS.get(device_type.lower(), _GENERIC)()

    def validate_parameters(self, device_type: str, device_params: Dict[str, Any]) -> Dict[str, Any]:
        """device_factory.py:196-227: defaults of the type overlaid with the given values; only the STT-MRAM validator
        checks anything (:229-246), the other types' validators are empty in the reference."""
        device_type = device_type.lower()
        out = dict(self.get_default_parameters(device_type))
        out.update(device_params)
        if device_type == "stt_mram":
            for key in ("volume", "saturation_magnetization", "damping", "uniaxial_anisotropy", "polarization"):
                if key not in out:
                    raise ValueError(f"Missing required parameter for STT-MRAM: {key}")
            if out["volume"] <= 0:
                raise ValueError("Volume must be positive")
            if out["saturation_magnetization"] <= 0:
                raise ValueError("Saturation magnetization must be positive")
            if not 0 <= out["damping"] <= 1:
                raise ValueError("Damping must be between 0 and 1")
            if not 0 <= out["polarization"] <= 1:
                raise ValueError("Polarization must be between 0 and 1")
        return out

    def create_default_device(self, device_type: str) -> BaseSpintronicDevice:
        return self.create_device(device_type, self.get_default_parameters(device_type))


def params_valid_as_stt(d: Dict[str, Any]) -> bool:
    """Outcome of ``validate_parameters(device_params)`` as RobustLLGSSolver calls it, i.e. with the default
    ``device_type='stt_mram'`` whatever the device is (utils/validation.py:176-234,491-493;
    utils/robust_solver.py:179).  False -> every solve falls back and the step is a no-op on m."""
    def positive(v, lo):
        try:
            v = float(v)
        except (TypeError, ValueError):
            return False
        return bool(np.isfinite(v) and v > 0 and v >= lo)

    def probability(v):
        try:
            v = float(v)
        except (TypeError, ValueError):
            return False
        return bool(np.isfinite(v) and 0 <= v <= 1)

    for key in ("volume", "saturation_magnetization", "damping", "uniaxial_anisotropy", "easy_axis", "polarization"):
        if key not in d:
            return False
    try:
        e = np.array(d["easy_axis"], dtype=float)
    except (TypeError, ValueError):
        return False
    if e.shape != (3,) or not np.all(np.isfinite(e)) or np.linalg.norm(e) < 1e-12:
        return False
    return (positive(d["volume"], 1e-30) and positive(d["saturation_magnetization"], 1e3)
            and probability(d["damping"]) and positive(d["uniaxial_anisotropy"], 1e3)
            and probability(d["polarization"]))


def flatten_params(device: BaseSpintronicDevice) -> "_lib.StgDeviceParams":
    """One reference-style device -> the C-ABI record (include/spintorque_hip.h: stg_device_params), using the
    defaults of the reference's own .get() calls (simple_solver.py:126-131; llgs_solver.py:79-82,192-205;
    spin_torque_env.py:476,502)."""
    d = device.device_params
    p = _lib.StgDeviceParams()
    p.damping = d.get("damping", 0.01)
    p.ms = d.get("saturation_magnetization", 800e3)
    p.ku = d.get("uniaxial_anisotropy", 1e6)
    p.volume = d.get("volume", 1e-24)
    p.polarization = d.get("polarization", 0.7)
    p.easy_axis[:] = [float(x) for x in np.asarray(d.get("easy_axis", [0, 0, 1]), dtype=float)]
    p.demag[:] = [float(x) for x in np.asarray(d.get("demag_factors", [0, 0, 1]), dtype=float)]
    p.a_ex = d.get("exchange_constant", 20e-12)
    p.area = d.get("area", 1e-14)
    p.r_p = d.get("resistance_parallel", 1e3)
    p.r_ap = d.get("resistance_antiparallel", 2e3)
    p.ref_m[:] = [float(x) for x in np.asarray(d.get("reference_magnetization", [0, 0, 1]), dtype=float)]
    p.r_series = device.series_resistance() if isinstance(device, SOTMRAMDevice) else 0.0
    # coefficients of the opt-in device-physics torque model (stg_config.torque_model = 1)
    p.sot_tau_dl = p.sot_tau_fl = 0.0
    p.sot_sigma[:] = [0.0, 1.0, 0.0]
    p.vcma_xi, p.vcma_td, p.vcma_vbd = 0.0, 1e-9, 2.0
    if isinstance(device, SOTMRAMDevice):
        p.sot_tau_dl, p.sot_tau_fl = device.tau_dl_factor, device.tau_fl_factor
        j_hat = np.asarray(d.get("current_direction", [1.0, 0.0, 0.0]), dtype=float)
        j_hat = j_hat / np.linalg.norm(j_hat)
        p.sot_sigma[:] = [float(x) for x in np.cross(np.array([0.0, 0.0, 1.0]), j_hat)]      # sot_mram.py:183-186
    if isinstance(device, VCMAMRAMDevice):
        p.vcma_xi, p.vcma_td, p.vcma_vbd = device.vcma_coefficient, device.dielectric_thickness, device.breakdown_voltage
    p.shape_demag[:] = [0.0, 0.0, 0.0]
    if isinstance(device, (SOTMRAMDevice, VCMAMRAMDevice)):
        p.shape_demag[:] = [float(x) for x in device.demag_factors()]
    if device.device_type not in _lib.DEV_TYPES:
        raise ValueError(f"device type '{device.device_type}' is not supported on the GPU step path")
    p.dev_type = _lib.DEV_TYPES[device.device_type]
    p.params_valid = int(params_valid_as_stt(d))
    return p


# double-valued fields of stg_device_params in declaration order (= the rows of the per-env parameter block)
PARAM_ROWS = (["damping", "ms", "ku", "volume", "polarization"] + [f"easy_axis{k}" for k in range(3)]
              + [f"demag{k}" for k in range(3)] + ["a_ex", "area", "r_p", "r_ap"] + [f"ref_m{k}" for k in range(3)]
              + ["r_series", "sot_tau_dl", "sot_tau_fl"] + [f"sot_sigma{k}" for k in range(3)]
              + ["vcma_xi", "vcma_td", "vcma_vbd"] + [f"shape_demag{k}" for k in range(3)])
assert len(PARAM_ROWS) == _lib.STG_NPARAM

# reference dict key -> row name(s) of the per-env block that a per-env override of that key fills
_OVERRIDE_ROWS = {
    "damping": ["damping"], "saturation_magnetization": ["ms"], "uniaxial_anisotropy": ["ku"], "volume": ["volume"],
    "polarization": ["polarization"], "easy_axis": [f"easy_axis{k}" for k in range(3)],
    "demag_factors": [f"demag{k}" for k in range(3)], "exchange_constant": ["a_ex"], "area": ["area"],
    "resistance_parallel": ["r_p"], "resistance_antiparallel": ["r_ap"],
    "reference_magnetization": [f"ref_m{k}" for k in range(3)],
}


def per_env_param_block(base: "_lib.StgDeviceParams", n: int, overrides: Dict[str, Any]):
    """Per-env parameter block for `stg_set_params_per_env` (device-to-device variation / domain randomisation): every
    env starts from the flattened `base` record and the reference-style keys in `overrides` replace single fields with
    arrays of length n (vectors: shape [n, 3]).  Returns (block float64 [STG_NPARAM, n], dev_type uint8 [n], valid uint8
    [n]); `valid` is the per-env outcome of the reference's ``validate_parameters(..., 'stt_mram')`` gate
    (utils/validation.py:176-234) evaluated on the overridden values."""
    flat = {}
    for name, _ in base._fields_:
        v = getattr(base, name)
        if hasattr(v, "__len__"):
            for k in range(len(v)):
                flat[f"{name}{k}"] = float(v[k])
        else:
            flat[name] = v
    block = np.empty((_lib.STG_NPARAM, n), dtype=np.float64)
    for r, name in enumerate(PARAM_ROWS):
        block[r] = flat[name]
    for key, val in overrides.items():
        if key not in _OVERRIDE_ROWS:
            raise ValueError(f"per-env override of '{key}' is not supported (supported: {sorted(_OVERRIDE_ROWS)})")
        rows = _OVERRIDE_ROWS[key]
        arr = np.asarray(val, dtype=np.float64)
        if arr.shape != ((n,) if len(rows) == 1 else (n, 3)):
            raise ValueError(f"per-env '{key}' must have shape {(n,) if len(rows) == 1 else (n, 3)}, got {arr.shape}")
        for k, rname in enumerate(rows):
            block[PARAM_ROWS.index(rname)] = arr if len(rows) == 1 else arr[:, k]
    row = {name: block[r] for r, name in enumerate(PARAM_ROWS)}
    e = np.stack([row["easy_axis0"], row["easy_axis1"], row["easy_axis2"]], axis=1)
    with np.errstate(invalid="ignore"):
        valid = (np.isfinite(e).all(axis=1) & (np.linalg.norm(e, axis=1) >= 1e-12)
                 & np.isfinite(row["volume"]) & (row["volume"] >= 1e-30) & (row["volume"] > 0)
                 & np.isfinite(row["ms"]) & (row["ms"] >= 1e3)
                 & np.isfinite(row["damping"]) & (row["damping"] >= 0) & (row["damping"] <= 1)
                 & np.isfinite(row["ku"]) & (row["ku"] >= 1e3)
                 & np.isfinite(row["polarization"]) & (row["polarization"] >= 0) & (row["polarization"] <= 1))
    valid &= bool(base.params_valid)          # (a base dict that fails the gate, e.g. without 'polarization', stays a no-op)
    dev_type = np.full(n, int(base.dev_type), dtype=np.uint8)
    return block, dev_type, valid.astype(np.uint8)


def per_env_param_block_multi(bases, class_index, n: int, overrides: Dict[str, Any]):
    """As `per_env_param_block` for a mixed batch: env i starts from the flattened record `bases[class_index[i]]` (its device
    type's own fields -- series resistance, SOT factors, VCMA coefficients -- included), then the per-env `overrides` apply.
    Returns (block [STG_NPARAM, n], dev_type uint8 [n], valid uint8 [n])."""
    cls = np.zeros(n, dtype=np.int64) if class_index is None else np.asarray(class_index, dtype=np.int64).reshape(-1)
    if cls.shape != (n,) or cls.min() < 0 or cls.max() >= len(bases):
        raise ValueError("class_index must hold one index into the device list per env")
    if len(bases) == 1:
        return per_env_param_block(bases[0], n, overrides)
    parts = [per_env_param_block(b, n, overrides) for b in bases]
    block = np.empty_like(parts[0][0])
    dev_type = np.empty(n, dtype=np.uint8)
    valid = np.empty(n, dtype=np.uint8)
    for k, (bk, tk, vk) in enumerate(parts):
        sel = cls == k
        block[:, sel] = bk[:, sel]
        dev_type[sel] = tk[sel]
        valid[sel] = vk[sel]
    return block, dev_type, valid
