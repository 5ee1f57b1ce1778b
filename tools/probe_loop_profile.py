"""Experiment: per-segment cycle counts of the RK45 attempt loop (library built with -DSTG_PROFILE_LOOP)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import torch
import spin_torque_gym_amd as stg
from spin_torque_gym_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
thermal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
random_pulses = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False   # 1: U[0.1, 1] ns with the default lane schedule
fac = stg.DeviceFactory()
par = fac.get_default_parameters("stt_mram"); par.update(volume=9.7e-6)
env = stg.SpinTorqueVecEnv(n, solver="rk45", include_thermal_fluctuations=thermal, device_params=par, seed=1, lane_sort=None if random_pulses else False)
env.reset(seed=0)
a = torch.zeros((n, 2), dtype=torch.float32); a[:, 0] = 1e6; a[:, 1] = 1e-9
if random_pulses:
    a[:, 1] = torch.rand(n, generator=torch.Generator().manual_seed(3)) * 0.9e-9 + 0.1e-9
a = a.cuda()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
env.step(a)
e0.record()
env.step(a)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
lib = _lib.load()
out = (ctypes.c_longlong * 16)()
lib.stg_debug_prof.restype = ctypes.c_int
assert lib.stg_debug_prof(out) == 0
att = out[10]
names = ["loop top/guard", "stage 2", "stage 3", "stage 4", "stage 5", "stage 6", "y_new + f_new", "error norm + controller"]
tot = sum(out[k] for k in range(8))
print(f"n={n} thermal={thermal} random={random_pulses}: step {ms:.3f} ms; wavefront 0 of workgroup 0: attempts {att}, total {tot} ticks = {tot/att:.0f} per attempt; {ms*1e6/max(tot,1):.3f} ns per tick if that wavefront spans the launch")
print(f"  integrator found the next chunk missing {out[11]} times ({out[12]} polls)")
for k, nm in enumerate(names):
    print(f"  {nm:26s} {out[k]/att:8.1f} ticks/attempt")
