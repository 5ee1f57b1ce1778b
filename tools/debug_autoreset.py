import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'tests'), os.path.join(ROOT,'spin-torque-rl-gym_amd')]
import numpy as np, torch
import spin_torque_gym_amd as stg
from conftest import stt_default_params
import test_gpu_parity as T
outs = T._run_pair(stg, 256, 5, T._uniform_actions(2e6, 1e-10, 3e-10), device_params=stt_default_params(volume=8.75e-11),
                     include_thermal_fluctuations=False, solver="rk4", max_steps=2, autoreset=True, seed=21)
hip, ora = outs
for s in range(1, len(hip)):
    d=np.abs(hip[s]['m']-ora[s]['m']).max(axis=0)
    bad=np.nonzero(d>1e-5)[0]
    print('step',s,'max',d.max(),'n>1e-5',len(bad),'trunc',hip[s]['trunc'].sum(),'term',hip[s]['term'].sum(), 'first bad',bad[:5], 'status eq', np.array_equal(hip[s]['status'],ora[s]['status']))
    if len(bad):
        i=bad[0]; print('   hip m',hip[s]['m'][:,i],'ora m',ora[s]['m'][:,i],'term/trunc prev', hip[s-1].get('term',[0]*256)[i] if s>1 else None, hip[s]['term'][i],hip[s]['trunc'][i], 'obs hip',hip[s]['obs'][i][:6],'obs ora',ora[s]['obs'][i][:6])
