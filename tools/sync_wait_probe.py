#!/usr/bin/env python3
"""Round 3 probe: where does the sporadic 25-85 ms host-side stall of a timed block come from?

The hip-trace of round 3 (profiles/r03_stall_trace_excerpt.txt) shows one thread, no Python collector pass and no HIP
call between the last enqueue and the fence: `hipDeviceSynchronize` itself returns tens of milliseconds after the last
kernel of the block has ended.  ROCclr waits for a completion signal by spinning for 100 us and then sleeping in the
kernel driver until the completion interrupt wakes it.  This probe runs the same short block of step kernels many
times with three fences and reports how late each returns with respect to the device's own span:

  sync   torch.cuda.synchronize()                  -> hipDeviceSynchronize   (spin 100 us, then interrupt-driven sleep)
  evsync event.synchronize()                       -> hipEventSynchronize    (same wait path)
  spin   while not event.query(): pass             -> hipEventQuery polling  (never sleeps)

    python3 tools/sync_wait_probe.py [--blocks 1500] [--envs 4096] [--steps 5] [--spin-flag]

--spin-flag calls hipSetDeviceFlags(hipDeviceScheduleSpin) through libamdhip64 before anything else touches the GPU,
which makes the runtime's own waits active (the `sync` fence then never sleeps either).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spin-torque-rl-gym_amd")):
    sys.path.insert(0, p)


def cgroup_cpu():
    """CPU bandwidth control of this container: quota (cpu.max) and the throttling statistics (cpu.stat)."""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            out["max"] = open(path).read().strip()
            break
        except OSError:
            pass
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            out["stat"] = {k: int(v) for k, v in (ln.split() for ln in open(path).read().splitlines())}
            break
        except (OSError, ValueError):
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=1500)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--spin-flag", action="store_true")
    ap.add_argument("--idle-ms", type=float, default=0.0, help="host sleep between blocks (a learner doing CPU work)")
    ap.add_argument("--cpu-burst", type=int, default=0, help="before every block: torch.rand of this many elements on the CPU "
                    "(multi-threaded through OpenMP, like the input generation of a benchmark row)")
    ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads (0 = leave the default)")
    args = ap.parse_args()
    flag_rc = None
    if args.spin_flag:
        hip = ctypes.CDLL("libamdhip64.so")
        flag_rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))          # hipDeviceScheduleSpin
    import torch
    import spin_torque_gym_amd as stg
    p = stg.DeviceFactory().get_default_parameters("stt_mram")
    p["volume"] = 8.75e-11
    if args.threads:
        torch.set_num_threads(args.threads)
    print("host: os.cpu_count %s, affinity %d, torch threads %d, cgroup %s" % (os.cpu_count(), len(os.sched_getaffinity(0)),
          torch.get_num_threads(), cgroup_cpu()), flush=True)
    env = stg.SpinTorqueVecEnv(args.envs, device_params=p, include_thermal_fluctuations=False, solver="rk4", seed=1, autoreset=True)
    env.reset(seed=0)
    b = env.backend
    dev = b.device
    g = torch.Generator().manual_seed(0)
    a = torch.empty((2, args.envs))
    a[0] = (torch.rand(args.envs, generator=g) * 2 - 1) * 2e6
    a[1] = 1e-10 + torch.rand(args.envs, generator=g) * 9e-10
    a = a.to(dev)
    for _ in range(3):
        b.step(a, autoreset=True)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {"args": vars(args), "hipSetDeviceFlags_rc": flag_rc, "fences": {}}
    for fence in ("sync", "evsync", "spin", "sync"):
        late = []
        cg0 = cgroup_cpu()
        for _ in range(args.blocks):
            if args.cpu_burst:
                torch.rand(args.cpu_burst)
            if args.idle_ms:
                time.sleep(args.idle_ms * 1e-3)
            t0 = time.perf_counter()
            ev0.record()
            for _ in range(args.steps):
                b.step(a, autoreset=True)
            ev1.record()
            if fence == "sync":
                torch.cuda.synchronize(dev)
            elif fence == "evsync":
                ev1.synchronize()
            else:
                while not ev1.query():
                    pass
            wall = (time.perf_counter() - t0) * 1e3
            late.append(wall - ev0.elapsed_time(ev1))
        late.sort()
        key = fence if fence not in out["fences"] else fence + "_again"
        out["fences"][key] = {"blocks": len(late), "late_ms_median": round(late[len(late) // 2], 4), "late_ms_p99": round(late[int(len(late) * 0.99)], 4),
                              "late_ms_max": round(late[-1], 3), "n_late_over_2ms": sum(x > 2.0 for x in late),
                              "n_late_over_20ms": sum(x > 20.0 for x in late), "worst5_ms": [round(x, 2) for x in late[-5:]]}
        cg1 = cgroup_cpu()
        out["fences"][key]["cgroup_delta"] = {k: cg1["stat"].get(k, 0) - cg0["stat"].get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec")} if cg0.get("stat") else None
        print(key, json.dumps(out["fences"][key]), flush=True)
    env.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
