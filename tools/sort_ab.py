import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'spin-torque-rl-gym_amd')]
import torch, numpy as np
import spin_torque_gym_amd as stg
from spin_torque_gym_amd.backend import EnvConfig, HipBackend
import bench
ns=[int(x) for x in sys.argv[1].split(',')] if len(sys.argv)>1 else [65536,131072,262144,524288]
for n in ns:
    for solver, thermal in (("rk4",0),("rk45",1)):
        for sort in (True, False, True):
            cfg=EnvConfig(solver=solver, include_thermal_fluctuations=bool(thermal), seed=1, lane_sort=sort)
            b=HipBackend(n,cfg); b.set_params([stg.flatten_params(stg.DeviceFactory().create_device('stt_mram', bench.stt_params(bench.volume_for(solver))))])
            b.reset(None,None,None,3)
            acts=bench.make_actions(6,n,b.device,5)
            ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(6)]
            for k in range(6):
                ev[k][0].record(); b.step(acts[k],autoreset=True); ev[k][1].record()
            torch.cuda.synchronize()
            ms=[a.elapsed_time(c) for a,c in ev]
            print(f"N={n} {solver} th={thermal} sort={int(sort)}: "+" ".join(f"{x:.3f}" for x in ms), b.counters())
            b.close()
