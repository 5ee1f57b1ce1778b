"""cfg4 device-physics row (262 144 mixed envs, RK4): order of a tile's workgroups.  Knobs are read at stg_create / per launch from the
environment, so one process walks the variants.  Kernel ms + the timeline's busy share / tail."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

bench.cap_host_threads()
bench.DEFAULT_BLOCKS = 3
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [262144]
thermal = int(sys.argv[2]) if len(sys.argv) > 2 else 0
VARIANTS = [("shipped", {}), ("regroup by first env's duration", dict(STG_REGROUP=1)), ("regroup by longest block's cost", dict(STG_REGROUP=2)),
            ("regroup by longest block's duration", dict(STG_REGROUP=3)),
            ("cost + snake on 3 rounds", dict(STG_REGROUP=2, STG_SNAKE=1, STG_SNAKE_ROUNDS=3)),
            ("cost + snake on 2 rounds", dict(STG_REGROUP=2, STG_SNAKE=1, STG_SNAKE_ROUNDS=2)),
            ("duration(longest) + snake on 3 rounds", dict(STG_REGROUP=3, STG_SNAKE=1, STG_SNAKE_ROUNDS=3)),
            ("shipped again", {})]
for n in sizes:
    for name, kn in VARIANTS:
        for k in ("STG_REGROUP", "STG_SNAKE", "STG_SNAKE_ROUNDS"):
            os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in kn.items()})
        m = bench.run_config(n, "rk4", thermal, 8, 2, 0, 1, 0, mixed=True, torque_model="device")
        pl = m["placement"][-1]
        print(f"n={n} thermal={thermal} {name}: kernel {m['kernel_ms_avg']:.4f} ms (min {m['kernel_ms_min']:.4f}); busy {pl['simd_busy_frac']}, tail {pl['last_simd_alone_frac']}", flush=True)
