import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spin-torque-rl-gym_amd"))
import torch
import spin_torque_gym_amd as stg
n = int(sys.argv[1]); solver = sys.argv[2]; K = int(sys.argv[3])
env = stg.SpinTorqueVecEnv(n, solver=solver, include_thermal_fluctuations=False, seed=1, autoreset=True, lane_sort=False)
env.reset(seed=0)
a = torch.zeros((K, 2, n), dtype=torch.float32, device="cuda"); a[:, 1] = 1e-12
for _ in range(12):
    env.step_many(a, actions_soa=True, out_every=False)
torch.cuda.synchronize()
