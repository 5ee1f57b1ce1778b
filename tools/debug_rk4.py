import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'tests'), os.path.join(ROOT,'spin-torque-rl-gym_amd')]
import numpy as np, torch
import spin_torque_gym_amd as stg
from spin_torque_gym_amd.backend import EnvConfig, HipBackend
from conftest import stt_default_params
import oracle
g2=np.load(os.path.join(ROOT,'tests/golden/G2_simple_rk4_stt.npz'))
def flat(d): return stg.flatten_params(stg.DeviceFactory().create_device('stt_mram', d))
table=[flat(stt_default_params(volume=8.75e-11)), flat(stt_default_params(volume=2e-11))]
n=len(g2['T'])
m0=np.array([g2['m0'][i] for i in g2['m0_index']])
cls=torch.tensor([0 if v>5e-11 else 1 for v in g2['volume']],dtype=torch.uint8)
b=HipBackend(n, EnvConfig(solver='rk4', include_thermal_fluctuations=False)); b.set_params(table, cls)
out=b.solve(torch.tensor(m0.T.copy()), torch.tensor(g2['J']), torch.tensor(g2['T']))
mf=out['m_final'].cpu().numpy().T
err=np.abs(mf-g2['m_final']).max(axis=1)
order=np.argsort(-err)[:15]
for k in order:
    print(k, 'err=%.3e'%err[k], 'vol=%g J=%g T=%.17g m0idx=%d n=%d'%(g2['volume'][k],g2['J'][k],g2['T'][k],g2['m0_index'][k],g2['n_steps'][k]), mf[k], g2['m_final'][k])
print('count > 1e-10:', (err>1e-10).sum(), 'of', n)
# H4 check: which cases have last-stage J switched off
for k in order[:15]:
    T=g2['T'][k]; nst=g2['n_steps'][k]; dt=T/nst
    print(k, 'last stage t+dt <= T ?', ((nst-1)*dt+dt)<=T, ' t+dt/2<=T', ((nst-1)*dt+dt/2)<=T)
