for lib in ${LIBS:-build/lib_base.so spin-torque-rl-gym_amd/spin_torque_gym_amd/libspintorque_hip.so build/lib_base.so spin-torque-rl-gym_amd/spin_torque_gym_amd/libspintorque_hip.so}; do
 for n in 4096 65536; do
  STG_HIP_LIBRARY=$PWD/$lib python3 bench.py --steps 6 --warmup 1 --cpu-baseline 0 --also 0 --solver euler --thermal 1 --envs-per-gpu $n 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $lib) euler n=$n: kernel %.3f ms' % d['roofline']['kernel_ms_avg'])"
  STG_HIP_LIBRARY=$PWD/$lib python3 bench.py --steps 6 --warmup 1 --cpu-baseline 0 --also 0 --solver rk4 --thermal 1 --envs-per-gpu $n 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $lib) rk4 n=$n: kernel %.3f ms' % d['roofline']['kernel_ms_avg'])"
 done
done
