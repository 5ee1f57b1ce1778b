#!/bin/bash
# mixed STT/SOT/VCMA batch with the thermal field on at 65 536 envs (wave-specialised launch + class table in LDS):
# kernel ms per library.  usage: LIBS="a.so b.so" tools/mixed_thermal_ab.sh
for lib in $LIBS; do
 for solver in rk45 rk4; do
  STG_HIP_LIBRARY=$PWD/$lib python3 -c "
import bench
m = bench.run_config(65536, '$solver', 1, 6, 1, 0, 1, 0, mixed=True)
print('$(basename $lib) $solver mixed thermal 65536: kernel %.3f ms' % m['kernel_ms_avg'])" 2>/dev/null
 done
done
