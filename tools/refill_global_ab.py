"""Global-queue lane refill (STG_REFILL_GLOBAL=1) against the shipped schedule, RK45 + thermal, kernel ms; knobs are read at stg_create,
so one process walks through the variants.  Also checks every variant's records against the shipped ones bit for bit (one step)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench  # noqa: E402
import spin_torque_gym_amd as stg  # noqa: E402

bench.cap_host_threads()
bench.DEFAULT_BLOCKS = 3
thermal = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [98304, 131072, 196608, 262144, 524288, 1048576]


def setenv(**kw):
    for k in ("STG_REFILL", "STG_REFILL_GLOBAL"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in kw.items()})


def records(n):
    env = stg.SpinTorqueVecEnv(n, device_params=bench.stt_params(9.7e-6), solver="rk45", include_thermal_fluctuations=bool(thermal), seed=7, autoreset=True)
    env.reset(seed=3)
    a = bench.make_actions(2, n, env.backend.device, 11)
    for k in range(2):
        env.backend.step(a[k], autoreset=True)
    r = env.backend.packed.clone()
    c = env.backend.counters()
    env.close()
    return r, c


for n in sizes:
    nblk = (n + 4095) // 4096 * 64
    setenv()
    ref, cref = records(n)
    m = bench.run_config(n, "rk45", thermal, 6, 2, 0, 1, 0)
    print(f"n={n} thermal={thermal} shipped: kernel {m['kernel_ms_avg']:.3f} ms (min {m['kernel_ms_min']:.3f})", flush=True)
    for nw in (1024, 2048):
        if nblk <= nw:
            continue
        r = (nblk + nw - 1) // nw
        for chk in (64, 32):
            setenv(STG_REFILL=f"{r},{chk}", STG_REFILL_GLOBAL=1)
            got, c = records(n)
            same = bool(torch.equal(got, ref)) and c == cref
            m = bench.run_config(n, "rk45", thermal, 6, 2, 0, 1, 0)
            pl = m["placement"][-1]
            print(f"n={n} thermal={thermal} global queue, {nw} wavefronts (check every {chk}): kernel {m['kernel_ms_avg']:.3f} ms (min {m['kernel_ms_min']:.3f}); "
                  f"bit-identical {same}; busy {pl['simd_busy_frac']}, tail {pl['last_simd_alone_frac']}", flush=True)
        setenv(STG_REFILL=f"{r},64")
        m = bench.run_config(n, "rk45", thermal, 6, 2, 0, 1, 0)
        print(f"n={n} thermal={thermal} fixed queues, {nw} wavefronts: kernel {m['kernel_ms_avg']:.3f} ms (min {m['kernel_ms_min']:.3f})", flush=True)
