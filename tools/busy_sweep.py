"""Size sweep with the placement table's busy share and tail: finds launch sizes whose schedule leaves SIMDs idle.
python3 tools/busy_sweep.py <solver> <thermal> [mixed: 0 | ref | device] [sizes,comma]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

bench.cap_host_threads()
bench.DEFAULT_BLOCKS = 1
solver, thermal = sys.argv[1], int(sys.argv[2])
mixed = sys.argv[3] if len(sys.argv) > 3 else "0"
sizes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [16384, 32768, 49152, 65536, 70000, 81920, 90112, 98304, 114688, 131072, 150000,
                                                                             163840, 196608, 229376, 262144, 327680, 393216, 524288, 786432, 1048576]
for n in sizes:
    m = bench.run_config(n, solver, thermal, 6, 2, 0, 1, 0, mixed=mixed != "0", torque_model="device" if mixed == "device" else "reference", retime=False)
    pl = m["placement"][-1]
    print(f"{solver} thermal={thermal} mixed={mixed} n={n:8d}: kernel {m['kernel_ms_avg']:8.4f} ms  {n / m['kernel_ms_avg'] * 1e-6:7.1f} Menv-steps/s  wg {pl['workgroups']}x{pl['waves_per_workgroup']} "
          f"busy {pl['simd_busy_frac']}  tail {pl['last_simd_alone_frac']}  waves/SIMD {pl['integrating_per_simd']}", flush=True)
