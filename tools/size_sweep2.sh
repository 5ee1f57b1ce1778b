#!/bin/bash
# Kernel time against batch size across the schedule's regime boundaries (round 3): tools/size_sweep2.sh > out.txt
LIB=${LIB:-spin-torque-rl-gym_amd/spin_torque_gym_amd/libspintorque_hip.so}
SIZES="4096 10000 32768 36864 45000 50000 60000 65536 66000 70000 81920 90000 100000 110000 125000 131072 132000 150000 165000 170000 190000 196608 230000 262144 300000 400000 524288 1048576"
for solver in rk45 rk4; do for t in 1 0; do
  cfgs=(); for n in $SIZES; do cfgs+=("$n $t"); done
  SOLVER=$solver LIBS=$LIB bash tools/ab_sizes.sh 1 "${cfgs[@]}"
done; done
