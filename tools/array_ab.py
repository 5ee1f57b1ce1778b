"""A/B of the array-env kernel variants (STG_ARRAY_VARIANT: 0 generic, 1 unrolled 4x4 = default):
python3 tools/array_ab.py [n]  -- kernel ms per launch, 262144 4x4 arrays by default, modes global / row / column."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
bench.cap_host_threads()
for rep in range(2):
    for mode in ("global", "row", "column"):
        for v in ((0, 2, 1) if mode == "global" else (0, 1)):   # (2: the 4 x 4 LDS kernel in global mode; 1 there: pattern in registers)
            os.environ["STG_ARRAY_VARIANT"] = str(v)
            r = bench.run_array_config(n, mode, 8, 0)
            print(f"rep {rep} {mode:7s} variant {v}: kernel {r['kernel_ms_avg']:.4f} ms  ({n / (r['wall_s'] / 8):.3e} array-steps/s)", flush=True)
