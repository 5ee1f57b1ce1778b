"""Thermal fixed-step kernels (inline normals) at throughput sizes -- kernel ms; run once per library build (STG_HIP_LIBRARY=...)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

bench.cap_host_threads()
tag = os.path.basename(os.environ.get("STG_HIP_LIBRARY", "in-tree"))
for n in (131072, 262144, 1048576):
    for solver in ("rk4", "euler"):
        m = bench.run_config(n, solver, 1, 8, 2, 0, 1, 0)
        print(f"[{tag}] {solver} thermal n={n}: kernel {m['kernel_ms_avg']:.4f} ms", flush=True)
