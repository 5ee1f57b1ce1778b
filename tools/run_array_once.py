"""One array-env configuration for profiling: python3 tools/run_array_once.py <mode> <rows> <cols> [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

mode, rows, cols = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 262144
r = bench.run_array_config(n, mode, 5, 0, size=(rows, cols))
print(f"{mode} {rows}x{cols} n={n}: {r['value']:.3e} array-steps/s, {r['roofline']['achieved']:.0f} GB/s, kernel {r['roofline']['kernel_ms_avg']:.4f} ms")
