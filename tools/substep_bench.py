"""Workload 2a of SURVEY 8d: env-steps that are a single RK45 step of 1 ps -- the only regime near the HBM ridge."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'spin-torque-rl-gym_amd')]
import torch, numpy as np
import spin_torque_gym_amd as stg
from spin_torque_gym_amd.backend import EnvConfig, HipBackend
import bench
for n in (262144, 1048576, 4194304):
    for solver, sort in (("rk45", False), ("rk4", False)):
        cfg=EnvConfig(solver=solver, include_thermal_fluctuations=False, seed=1, lane_sort=sort, max_steps=1000000)
        b=HipBackend(n,cfg); b.set_params([stg.flatten_params(stg.DeviceFactory().create_device('stt_mram', bench.stt_params(bench.volume_for(solver))))])
        b.reset(None,None,None,3)
        a=torch.zeros((2,n),dtype=torch.float32,device=b.device); a[0]=1e6; a[1]=1e-12
        K=20
        ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for k in range(K):
            ev[k][0].record(); b.step(a,autoreset=False); ev[k][1].record()
        torch.cuda.synchronize()
        ms=np.median([x.elapsed_time(y) for x,y in ev][5:])
        c=b.counters()
        print(f"N={n} {solver}: {ms*1e3:.1f} us/step  {n/ms*1e3:.3e} env-steps/s  HBM {160*n/ms/1e6:.1f} GB/s ({160*n/ms/1e6/8000*100:.1f}% of 8 TB/s)  work/env-step {c['work_units']/c['env_steps']:.2f}")
        b.close()
