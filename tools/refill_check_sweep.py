"""Lane refill: attempts between refill points (and, optionally, the wavefront count) over sizes; the library reads
STG_REFILL="<envs per lane>,<check>" when a context is created.
usage: python3 tools/refill_check_sweep.py <check> [waves] [thermal]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
bench.cap_host_threads(); bench.DEFAULT_BLOCKS = 3
chk = int(sys.argv[1])
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
th = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for n in (131072, 262144, 524288, 1048576):
    nblk = n // 64
    nw = waves or (1024 if nblk <= 8 * 1024 else 2048)
    r = max(2, (nblk + nw - 1) // nw)
    if chk > 0:
        os.environ["STG_REFILL"] = f"{r},{chk}"
    m = bench.run_config(n, "rk45", th, 6, 2, 0, 1, 0)
    pl = m["placement"][-1]
    print(f"[check={chk or 'default'} waves={nw if chk > 0 else 'auto'}] rk45 thermal={th} n={n}: kernel {m['kernel_ms_avg']:.3f} ms (min {m['kernel_ms_min']:.3f}) wg {pl['workgroups']}x{pl['waves_per_workgroup']}", flush=True)
