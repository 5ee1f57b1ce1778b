"""A/B of the number of RK4 sub-steps per chunk of the producer/consumer handshake (STG_HIP_LIBRARY=<variant>): kernel ms of the
fixed-step thermal launches.  usage: [STG_HIP_LIBRARY=build/lib_rk4c2.so] python3 tools/rk4_chunk_ab.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
bench.cap_host_threads(); bench.DEFAULT_BLOCKS = 3
tag = os.path.basename(os.environ.get("STG_HIP_LIBRARY", "shipped"))
for solver, n in (("rk4", 32768), ("rk4", 65536), ("rk4", 81920), ("euler", 65536)):
    m = bench.run_config(n, solver, 1, 8, 2, 0, 1, 0)
    print(f"[{tag}] {solver} thermal n={n}: kernel {m['kernel_ms_avg']:.4f} ms (min {m['kernel_ms_min']:.4f})", flush=True)
