import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'tests'), os.path.join(ROOT,'spin-torque-rl-gym_amd')]
import numpy as np, torch
import spin_torque_gym_amd as stg
from spin_torque_gym_amd.backend import EnvConfig, HipBackend
from conftest import stt_default_params
from helpers import device_reset_draw
cfg=EnvConfig(solver='rk4', include_thermal_fluctuations=False, seed=21, max_steps=1)
b=HipBackend(4,cfg); b.set_params([stg.flatten_params(stg.DeviceFactory().create_device('stt_mram', stt_default_params()))])
b.reset(None,None,None,777)
a=torch.zeros((2,4)); a[1]=1e-10
b.step(a, autoreset=True)
st=b.get_state()
print(st['m'].cpu().numpy().T, st['rng_step'].tolist(), st['step_count'].tolist())
for seed in (21, 777):
  for rs in (0,1):
    print(seed, rs, device_reset_draw(seed, 0, rs, [[0,0,1.],[0,0,-1.]])[0])
