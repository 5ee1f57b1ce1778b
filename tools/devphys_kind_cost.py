"""Cost of one RK4 sub-step per device kind under the device-physics torque model: kernel ms of homogeneous 262 144-env batches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import torch
import bench
import spin_torque_gym_amd as stg
bench.cap_host_threads()
n = 262144
fac = stg.DeviceFactory()
acts = bench.make_actions(4, n, torch.device("cuda", 0), 7)
for kind in ("stt_mram", "sot_mram", "vcma_mram"):
    for tm in ("reference", "device"):
        p = fac.get_default_parameters(kind); p.update(polarization=0.7, volume=8.75e-11)
        env = stg.SpinTorqueVecEnv(n, device_type=kind, device_params=p, include_thermal_fluctuations=False, solver="rk4", seed=1, autoreset=True,
                                   torque_model=tm)
        env.reset(seed=0)
        b = env.backend
        for k in range(2): b.step(acts[k], autoreset=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(8): b.step(acts[k % 4], autoreset=True)
        e1.record(); torch.cuda.synchronize()
        c = b.counters()
        print(f"{kind} torque_model={tm}: {e0.elapsed_time(e1)/8:.4f} ms/step, noop steps {c['noop_steps']}", flush=True)
        env.close()
