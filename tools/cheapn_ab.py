"""Ceiling of a cheaper normal generator: kernel ms of the thermal rows, once per library build
(STG_HIP_LIBRARY=build/lib_cheapn.so is the experiment build with the ~9.5-slot stand-in, wrong distribution; unset = shipped).
Run both alternately inside ONE gpurun call: tools/cheapn_ab.sh."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

bench.cap_host_threads()
bench.DEFAULT_BLOCKS = 3
tag = os.path.basename(os.environ.get("STG_HIP_LIBRARY", "shipped"))
for name, n, solver in (("RK4 + thermal 65536 (producer-bound pairs)", 65536, "rk4"), ("RK45 + thermal 65536 (headline)", 65536, "rk45"),
                        ("RK45 + thermal 131072 (cfg5 shard, inline normals)", 131072, "rk45"), ("RK4 + thermal 262144 (inline)", 262144, "rk4"),
                        ("RK45 + thermal 1048576 (refill, inline)", 1048576, "rk45")):
    m = bench.run_config(n, solver, 1, 8, 2, 0, 1, 0)
    print(f"[{tag}] {name}: kernel {m['kernel_ms_avg']:.4f} ms (min {m['kernel_ms_min']:.4f}), work units/env-step {m['work_units'] / max(m['env_steps'], 1):.1f}", flush=True)
