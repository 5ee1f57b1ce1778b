#!/usr/bin/env python3
"""bench.py -- env-steps/s of the SpinTorque-v0 step path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of every environment of the batch: one launch of the fused step kernel per GPU (action
clamp -> LLGS integration over the pulse -> energy -> observation -> reward -> termination), and for N > 1 the single
all-gather of (obs, reward, terminated, truncated).  Inputs (states, the [K,2,N] action tensor) are resident in HBM
before the timed region.  Weak scaling: --envs-per-gpu environments per GPU whatever N is.

Headline workload (config.workload): BASELINE.json configs[2] / the north-star target -- 65 536 STT-MRAM environments
per GPU, thermal field on at 300 K (in-kernel Philox), LLGSSolver semantics (SciPy RK45, rtol 1e-6, atol 1e-9,
max_step 1 ps), random pulses J ~ U[-2e6, 2e6] A/m^2, duration ~ U[0.1, 1] ns (float32), `volume` rescaled to 9.7e-6 so
that the Slonczewski term is well conditioned for RK45 (SURVEY.md headline 3: at the default volume any J != 0 makes
the reference's solver diverge), device-side auto-reset of finished episodes.  `also` carries the other
configurations (cfg2: 4096 envs T = 0 K; the env's own RK4 solver; cfg4: 262 144 mixed STT/SOT/VCMA envs; cfg2a; the
array env).  Every configuration timed here is compared with the oracle at the same size in tests/test_gpu_fullsize.py.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.

roofline.  The binding roof of a full env-step is fp64 VALU, not HBM.  `achieved` is
the EXECUTED fp64 rate -- 64 lanes x (2 SQ_INSTS_VALU_FMA_F64 + _MUL_F64 + _ADD_F64) per launch, from the hardware
counters, over the kernel's HIP-event time -- so `frac` <= 1 by construction; `valu_issue_frac` is the share of the
SIMDs' VALU issue slots the launch used; the reference formulation's flop count (305 per RK4 sub-step, 745
per RK45 attempt -- more than the kernels execute, because they fold constants) is kept apart as `work_equiv`.
`traffic` is HBM bytes per launch from FETCH_SIZE (x2: gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE.
The counters are collected LIVE: at N = 1 this script first runs itself three times under `rocprofv3 --kernel-trace
--pmc ...` (one pass per counter group: FETCH_SIZE and WRITE_SIZE cannot share one) on the same seeded workloads,
BEFORE this process touches the GPU, and maps the dispatches to the rows through marker launches.  If rocprofv3 is not
usable, the committed table profiles/r02_pmc_rows.json is used when it was measured on this very library build (sha256
of the .so), and `frac`/`traffic` are null otherwise: no stale number is ever printed.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "spin-torque-rl-gym_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# algorithmic work per unit
BYTES_PER_ENV_STEP = 160          # homogeneous params: 68 B read + 90 B written (+2 B rounding in the survey's figure)
BYTES_PER_ENV_STEP_MIXED = 161    # + 1 B class index
FLOPS_PER_RK4_SUBSTEP = 305       # reference formulation (work_equiv only)
FLOPS_PER_RK45_ATTEMPT = 745
PEAK_FP64_VALU_TFLOPS = 78.6      # MI355X vector fp64 (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 1024                     # 256 CUs x 4
NOMINAL_CLOCK_HZ = 2.4e9          # a wave64 VALU instruction occupies its SIMD for 4 cycles
PMC_PASSES = {
    "flops": "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}
PMC_TABLE = os.path.join(ROOT, "profiles", "r02_pmc_rows.json")
MAIN_KERNELS = ("stg_step_kernel", "stg_array_step_kernel", "stg_array_step_individual_kernel")
MARKER_KERNEL = "stg_normals_kernel"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--solver", default="rk45", choices=["rk4", "rk45", "euler"])
    ap.add_argument("--thermal", type=int, default=1)
    ap.add_argument("--also", type=int, default=1, help="run the secondary configurations too (N=1 only)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--lane-sort", default="auto", choices=["auto", "on", "off"],
                    help="duration-sorted lane schedule (auto = on)")
    ap.add_argument("--wave-spec", default="auto", choices=["auto", "on", "off"],
                    help="producer/consumer wavefront pairs for the thermal kernels (auto = on up to 65536 envs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank code path on a single GPU)")
    ap.add_argument("--gather-algo", default="all_gather", choices=["all_gather", "p2p"],
                    help="N > 1: RCCL all-gather (default) or the one-shot point-to-point exchange")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: collect the hardware counters live with rocprofv3 child passes (N = 1 only)")
    ap.add_argument("--pmc-dump", default=None, help="write the per-row counter table (JSON) here as well")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)     # internal: run the rows once, write a manifest
    return ap.parse_args()


def library_sha256():
    from spin_torque_gym_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def stt_params(volume):
    import spin_torque_gym_amd as stg
    p = stg.DeviceFactory().get_default_parameters("stt_mram")
    p["volume"] = volume
    return p


def make_actions(k, n, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.empty((k, 2, n), dtype=torch.float32)
    a[:, 0] = (torch.rand((k, n), generator=g) * 2 - 1) * 2e6
    a[:, 1] = 1e-10 + torch.rand((k, n), generator=g) * 9e-10
    return a.to(device)


def volume_for(solver):
    # regimes in which the current actually drives switching: RK4 8.75e-11, RK45 9.7e-6
    return 9.7e-6 if solver == "rk45" else 8.75e-11


def cgroup_cpu_stat():
    """CPU-bandwidth statistics of this container (cgroup v2 cpu.stat, v1 fallback): nr_periods, nr_throttled, throttled_usec."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = {k: int(v) for k, v in (ln.split() for ln in open(path).read().splitlines())}
            if "throttled_time" in d:                   # v1 reports nanoseconds
                d["throttled_usec"] = d["throttled_time"] // 1000
            return d
        except (OSError, ValueError):
            continue
    return {}


class Marker:
    """PMC child passes only: one launch of a kernel nothing else uses, between rows, so that the dispatches of the
    counter CSV can be attributed to rows without counting on launch counts."""

    def __init__(self, enabled, device_index):
        self.b = None
        if enabled:
            from spin_torque_gym_amd.backend import EnvConfig, HipBackend
            self.b = HipBackend(64, EnvConfig(), device_index)

    def mark(self):
        if self.b is not None:
            self.b.thermal_normals(0, 0, 1)
            torch.cuda.synchronize(self.b.device)


def run_config(n_local, solver, thermal, steps, warmup, rank, world, device_index, mixed=False, seed=1234, lane_sort=None,
               torque_model="reference", wave_spec=None, gather_algo="all_gather", retime=True):
    """Builds the env, runs warmup + timed steps, returns a dict of measurements (times are this rank's)."""
    import spin_torque_gym_amd as stg
    import torch.distributed as dist
    kw = dict(include_thermal_fluctuations=bool(thermal), temperature=300.0, solver=solver, seed=seed, autoreset=True,
              lane_sort=lane_sort, torque_model=torque_model, wave_spec=wave_spec)
    if mixed:
        fac = stg.DeviceFactory()
        sot = fac.get_default_parameters("sot_mram"); sot.update(polarization=0.7, volume=volume_for(solver))
        vc = fac.get_default_parameters("vcma_mram"); vc.update(polarization=0.7, volume=volume_for(solver))
        kw.update(device_type=["stt_mram", "sot_mram", "vcma_mram"], device_params=[stt_params(volume_for(solver)), sot, vc])
        cls_global = (torch.arange(n_local * world) % 3).to(torch.uint8)
    else:
        kw.update(device_params=stt_params(volume_for(solver)))
        cls_global = None
    if world > 1:
        from spin_torque_gym_amd.distributed import ShardedSpinTorqueVecEnv
        env = ShardedSpinTorqueVecEnv(n_local * world, device_index=device_index, class_index=cls_global,
                                      gather_algo=gather_algo, **kw)
        backend = env.local.backend
    else:
        env = stg.SpinTorqueVecEnv(n_local, device_index=device_index, class_index=cls_global, **kw)
        backend = env.backend
    dev = backend.device
    acts = make_actions(warmup + steps, n_local, dev, seed + 17 * rank)
    if world > 1:
        # communicator set-up (lazy in RCCL) must not land in the timed block whatever --warmup is: reset() gathers once
        env.reset(seed=seed + rank)
        torch.cuda.synchronize(dev)
    else:
        env.reset(seed=seed)

    def one_step(k, gather=True):
        """world > 1: the step kernel writes its 56-byte records into this rank's slice of a global record array; step
        k-1's in-place all-gather (own stream) runs under step k's kernel; gather_end hands out typed views -- what a
        learner consumes, no copies (spin_torque_gym_amd/distributed.py)."""
        if world == 1:
            backend.step(acts[k], autoreset=True)
            return
        env.step(acts[k], gather=False, actions_are_local=True, actions_soa=True)
        if gather:
            if env.gather_in_flight:
                env.gather_end()
            env.gather_begin()

    for k in range(warmup):
        one_step(k)
    if world > 1 and env.gather_in_flight:
        env.gather_end()
    torch.cuda.synchronize(dev)

    gc_log = []          # (generation, start offset in s from the block's t0, duration in s) of every collector pass

    def timed_block(gather=True):
        """EXACTLY `steps` steps between barrier + synchronize on both sides; also the same span seen from the device."""
        backend.counters(reset=True)
        debug = bool(os.environ.get("STG_BENCH_DEBUG"))
        host_t = []
        gc_t = [0.0]

        def _gc_cb(phase, info):
            if phase == "start":
                gc_t[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], gc_t[0], time.perf_counter() - gc_t[0]))
        if debug:
            import gc
            gc.callbacks.append(_gc_cb)
        cg0 = cgroup_cpu_stat()
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        fin = torch.cuda.Event(enable_timing=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            if debug:
                host_t.append(time.perf_counter())
            starts[k].record()
            if world == 1:
                backend.step(acts[warmup + k], autoreset=True)       # the step kernel, on torch's current stream
                ends[k].record()
            else:
                env.step(acts[warmup + k], gather=False, actions_are_local=True, actions_soa=True)
                ends[k].record()
                if gather:
                    if env.gather_in_flight:
                        env.gather_end()                             # step k-1's gather ran under step k's kernel
                    env.gather_begin()                               # the single collective of a step, on its own stream
        if world > 1 and gather:
            env.gather_end()                                         # (typed global views: obs [N,12], reward, flags)
        fin.record()
        t_enq = time.perf_counter()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        if debug:
            import gc
            gc.callbacks.remove(_gc_cb)
            cg1 = cgroup_cpu_stat()
            print("debug host: enqueue of %d steps %.3f ms, synchronize %.3f ms; gc passes in block: %s; cgroup cpu.stat delta over the block: %s" % (
                steps, (t_enq - t0) * 1e3, (t1 - t_enq) * 1e3,
                [(g, round((a - t0) * 1e3, 3), round(d * 1e3, 3)) for g, a, d in gc_log if a >= t0],
                {k: cg1.get(k, 0) - cg0.get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec")}), file=sys.stderr, flush=True)
        return t1 - t0, [s.elapsed_time(e) for s, e in zip(starts, ends)], starts[0].elapsed_time(fin) * 1e-3

    # The GPU boxes of this pool show a sporadic ~80 ms hiccup (the device finishes -- its own event timestamps are
    # back-to-back -- but the host's synchronize returns late; seen in any configuration, about once per process).  A
    # block whose wall time exceeds what the device itself measured for the same span by more than 25 % + 2 ms is
    # re-timed (at most twice); `blocks_timed` in the output says how many blocks were run.  Every block is exactly
    # `steps` steps and the reported time is always host wall-clock time of one whole block.
    blocks = 0
    while True:
        blocks += 1
        wall, kern_ms, dev_span = timed_block()
        hiccup = 1.0 if wall > 1.25 * dev_span + 2e-3 else 0.0
        if world > 1:
            flag = torch.tensor([hiccup], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            hiccup = float(flag.item())
        if os.environ.get("STG_BENCH_DEBUG"):
            print("debug run_config n=%d %s th=%s tm=%s: block %d wall %.3f ms, device span %.3f ms, kernel ms %s" % (
                n_local, solver, thermal, torque_model, blocks, wall * 1e3, dev_span * 1e3, [round(x, 3) for x in kern_ms]),
                file=sys.stderr, flush=True)
        if not hiccup or blocks >= 3 or not retime:
            break
    c = backend.counters()
    # SURVEY 8e "report both": a learner that is data-parallel over the same ranks needs no gather at all
    wall_ng = timed_block(gather=False)[0] if (world > 1 and retime) else None
    gather_only_ms = None
    if world > 1 and retime:
        # the collective by itself (nothing to hide under): `steps` exchanges of the last step's records back to back
        env.step(acts[warmup], gather=False, actions_are_local=True, actions_soa=True)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            env.gather_again()                       # (re-send the records of the step above)
            env.gather_begin()
            env.gather_end()
        torch.cuda.synchronize(dev)
        dist.barrier()
        gather_only_ms = (time.perf_counter() - t0) / steps * 1e3
    env.close()
    return dict(gather_only_ms=gather_only_ms, wall_no_gather_s=wall_ng, wall_s=wall, device_span_s=dev_span, blocks_timed=blocks, kernel_ms_avg=float(np.mean(kern_ms)), kernel_ms_min=float(np.min(kern_ms)),
                env_steps=c["env_steps"], work_units=c["work_units"], noop_steps=c["noop_steps"], launches=steps, warmup=warmup)


def run_array_config(n, mode, steps, device_index, size=(4, 4), retime=True):
    """SpinTorqueArray-v0: N independent R x C arrays, random actions; this kernel is HBM-shaped."""
    import spin_torque_gym_amd as stg
    env = stg.SpinTorqueArrayVecEnv(n, size, action_mode=mode, seed=3, device_index=device_index, max_steps=10**6,
                                    success_threshold=2.0)
    env.reset(seed=1)
    dev = env.backend.device
    g = torch.Generator(device="cpu").manual_seed(5)
    ndev = size[0] * size[1]
    a_dim = 2 if mode == "global" else 3
    acts = torch.empty((steps + 2, a_dim, n), dtype=torch.float32)
    if mode == "global":
        acts[:, 0] = (torch.rand((steps + 2, n), generator=g) * 2 - 1) * 2e6
        acts[:, 1] = (torch.rand((steps + 2, n), generator=g) * 2 - 1) * 2e6      # read as the current (reference quirk)
    else:
        acts[:, 0] = torch.rand((steps + 2, n), generator=g) * ndev
        acts[:, 1] = (torch.rand((steps + 2, n), generator=g) * 2 - 1) * 2e6
        acts[:, 2] = 1e-10 + torch.rand((steps + 2, n), generator=g) * 9e-10
    acts = acts.to(dev)
    for k in range(2):
        env.backend.step(acts[k])
    torch.cuda.synchronize(dev)
    for _ in range(3 if retime else 1):    # re-timed after a host-side hiccup, as in run_config
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            env.backend.step(acts[k + 2])
            ev[k][1].record()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        if wall <= 1.25 * ev[0][0].elapsed_time(ev[-1][1]) * 1e-3 + 2e-3:
            break
    ms = float(np.mean([x.elapsed_time(y) for x, y in ev]))
    env.close()
    affected = {"individual": 1, "row": size[1], "column": size[0], "global": ndev}[mode]
    # algorithmic bytes per array-step: pattern + target + state + action read; addressed cells, obs, reward, flags, state written
    b = (ndev * 24 * 2 + 12 + 4 * a_dim) + (affected * 24 + ndev * 24 + 4 + 2 + 12 + 16)
    return dict(kind="array", wall_s=wall, kernel_ms_avg=ms, launches=steps, warmup=2, n=n, bytes_per_unit=b,
                workload=f"SpinTorqueArray-v0: {n} arrays of {size[0]}x{size[1]} STT cells, action_mode={mode}, dipolar coupling",
                kernel="stg_array_step_individual_kernel" if mode == "individual" else "stg_array_step_kernel")


def run_short_pulse_config(n, steps, device_index, K=1, retime=True):
    """SURVEY 8d workload 2a: every pulse is ONE 1 ps DP5 step (rk45, T = 0 K, J = 0, default STT parameters) -- the
    HBM-shaped end of the env-step kernel (~1.2 attempts per env-step).  K > 1 fuses K env-steps per launch."""
    import spin_torque_gym_amd as stg
    env = stg.SpinTorqueVecEnv(n, solver="rk45", include_thermal_fluctuations=False, seed=1, autoreset=True,
                               device_index=device_index, lane_sort=False)
    env.reset(seed=0)
    b = env.backend
    a = torch.zeros((K, 2, n), dtype=torch.float32, device=b.device)
    a[:, 1] = 1e-12
    call = (lambda: b.step(a[0], autoreset=True)) if K == 1 else (lambda: b.step_many(a, out_every=False, autoreset=True))
    for _ in range(2):
        call()
    for _ in range(3 if retime else 1):    # re-timed after a host-side hiccup, as in run_config
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize(b.device)
        b.counters(reset=True)
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            call()
            ev[k][1].record()
        torch.cuda.synchronize(b.device)
        wall = time.perf_counter() - t0
        if wall <= 1.25 * ev[0][0].elapsed_time(ev[-1][1]) * 1e-3 + 2e-3:
            break
    ms = float(np.mean([x.elapsed_time(y) for x, y in ev]))
    c = b.counters()
    env.close()
    # state read and written once per launch, actions read K times, outputs written once (out_every = False)
    return dict(kind="short", wall_s=wall, kernel_ms_avg=ms, launches=steps, warmup=2, n=n, K=K,
                bytes_per_launch=BYTES_PER_ENV_STEP * n + 8 * n * (K - 1), env_steps=c["env_steps"], work_units=c["work_units"],
                noop_steps=c["noop_steps"],
                workload=f"cfg2a: {n} STT envs, T=0K, rk45, every pulse = one 1 ps DP5 step, {K} env-step(s) per launch")


# ----------------------------------------------------------------------------------------------------------------------
# the rows: headline first, then the secondary configurations (N = 1 only)
# ----------------------------------------------------------------------------------------------------------------------
def row_specs(args):
    """(key, runner) in execution order -- the same list drives the timed parent run and the PMC child passes."""
    lane_sort = {"auto": None, "on": True, "off": False}[args.lane_sort]
    wave_spec = {"auto": None, "on": True, "off": False}[args.wave_spec]
    st = max(3, args.steps // 2)
    rows = [("headline", dict(kind="step", n=args.envs_per_gpu, solver=args.solver, thermal=args.thermal, mixed=False,
                              tm="reference", steps=args.steps, warmup=args.warmup, lane_sort=lane_sort, wave_spec=wave_spec))]
    if not args.also:
        return rows
    for name, n, solver, thermal, mixed, tm in (
            ("cfg2: 4096 STT envs, T=0K, rk45", 4096, "rk45", 0, False, "reference"),
            ("cfg2: 4096 STT envs, T=0K, rk4 (the env's own solver)", 4096, "rk4", 0, False, "reference"),
            ("cfg3: 65536 STT envs, thermal on, rk4", 65536, "rk4", 1, False, "reference"),
            ("cfg5 shard: 131072 STT envs (1 048 576 over 8 GPUs), thermal on, rk45", 131072, "rk45", 1, False, "reference"),
            ("cfg4: 262144 mixed STT/SOT/VCMA envs (class table in LDS), T=0K, rk4, reference RHS for all types",
             262144, "rk4", 0, True, "reference"),
            ("cfg4: 262144 mixed STT/SOT/VCMA envs, T=0K, rk4, device-physics torque terms per type (opt-in)",
             262144, "rk4", 0, True, "device")):
        if solver == args.solver and n == args.envs_per_gpu and bool(thermal) == bool(args.thermal) and not mixed:
            continue
        rows.append((name, dict(kind="step", n=n, solver=solver, thermal=thermal, mixed=mixed, tm=tm, steps=st, warmup=2,
                                lane_sort=None, wave_spec=None)))
    for K in (1, 8):
        rows.append((f"cfg2a K={K}", dict(kind="short", n=1048576, K=K, steps=st)))
    for mode in ("individual", "global"):
        rows.append((f"array {mode}", dict(kind="array", n=262144, mode=mode, steps=st)))
    return rows


def run_row(spec, rank, world, local_rank, retime=True, gather_algo="all_gather"):
    if spec["kind"] == "step":
        m = run_config(spec["n"], spec["solver"], spec["thermal"], spec["steps"], spec["warmup"], rank, world, local_rank,
                       mixed=spec["mixed"], torque_model=spec["tm"], lane_sort=spec["lane_sort"], wave_spec=spec["wave_spec"],
                       gather_algo=gather_algo, retime=retime)
        m["kind"] = "step"
        return m
    if spec["kind"] == "short":
        return run_short_pulse_config(spec["n"], spec["steps"], local_rank, K=spec["K"], retime=retime)
    return run_array_config(spec["n"], spec["mode"], spec["steps"], local_rank, retime=retime)


# ----------------------------------------------------------------------------------------------------------------------
# hardware counters: rocprofv3 child passes (before this process touches the GPU), or the committed table
# ----------------------------------------------------------------------------------------------------------------------
def pmc_child(args):
    """Runs every row once (no re-timing, no CPU baseline) with a marker launch before each, writes the manifest."""
    mk = Marker(True, 0)
    manifest = []
    for key, spec in row_specs(args):
        mk.mark()
        m = run_row(spec, 0, 1, 0, retime=False)
        manifest.append({"key": key, "launches": m["launches"], "warmup": m["warmup"]})
    mk.mark()
    with open(args.pmc_child, "w") as f:
        json.dump(manifest, f)


def parse_pmc_csv(outdir, manifest):
    """-> {row key: {counter: mean per timed launch}}.  Dispatches in dispatch order; a marker launch opens each row."""
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows.extend(csv.DictReader(fh))
    if not rows:
        raise RuntimeError("no counter_collection.csv produced")
    per_dispatch = {}
    for r in rows:
        name = r.get("Kernel_Name", "")
        if MARKER_KERNEL not in name and not any(k in name for k in MAIN_KERNELS):
            continue
        did = r.get("Dispatch_Id", r.get("Dispatch_ID", r.get("dispatch_id")))
        d = per_dispatch.setdefault(int(did), {"name": name, "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seq = [per_dispatch[k] for k in sorted(per_dispatch)]
    groups, cur = [], None
    for d in seq:
        if MARKER_KERNEL in d["name"]:
            cur = []
            groups.append(cur)
        elif cur is not None:
            cur.append(d)
    groups = groups[:len(manifest)]
    if len(groups) != len(manifest):
        raise RuntimeError(f"{len(groups)} marker groups for {len(manifest)} rows")
    out = {}
    for g, m in zip(groups, manifest):
        if len(g) != m["launches"] + m["warmup"]:
            raise RuntimeError(f"row {m['key']}: {len(g)} dispatches, expected {m['launches'] + m['warmup']}")
        timed = g[m["warmup"]:]
        names = sorted({d["name"] for d in timed})
        ctrs = {}
        for d in timed:
            for k, v in d["c"].items():
                ctrs[k] = ctrs.get(k, 0.0) + v / len(timed)
        out[m["key"]] = {"kernel_name": names[0] if len(names) == 1 else names, "counters": ctrs}
    return out


def workload_key(args):
    """What the rows of a counter table depend on besides the library build."""
    return {"envs_per_gpu": args.envs_per_gpu, "solver": args.solver, "thermal": int(bool(args.thermal)), "also": int(bool(args.also)),
            "lane_sort": args.lane_sort, "wave_spec": args.wave_spec}


def collect_pmc_live(argv):
    """Three rocprofv3 passes over `python3 bench.py --pmc-child ...`; returns ({row: {counter: value}}, note)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        # this process is itself being profiled (its GPU is initialised by the profiler's preloaded library already, and a
        # profiler inside a profiler measures nothing useful): no child passes
        return None, "running under a profiler: no nested passes"
    if not os.path.exists("/dev/kfd"):
        return None, "no GPU device node"
    work = tempfile.mkdtemp(prefix="stg_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    merged, t0 = {}, time.time()
    try:
        for tag, ctrs in PMC_PASSES.items():
            outdir, mani = os.path.join(work, tag), os.path.join(work, tag + ".json")
            cmd = [exe, "--kernel-trace", "--pmc", *ctrs.split(), "--output-format", "csv", "-d", outdir, "--",
                   sys.executable, os.path.abspath(__file__), *argv, "--pmc-child", mani, "--cpu-baseline", "0"]
            # own process group, so that a pass that hangs is killed together with the program it profiles (nothing may be
            # left on the GPU when the timed runs start)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                log, _ = proc.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, 9)
                except OSError:
                    pass
                proc.wait()
                return None, f"pass '{tag}' timed out"
            if proc.returncode != 0 or not os.path.exists(mani):
                return None, f"pass '{tag}' failed (rc {proc.returncode}): {log.decode(errors='replace')[-300:]}"
            for key, v in parse_pmc_csv(outdir, json.load(open(mani))).items():
                e = merged.setdefault(key, {"kernel_name": v["kernel_name"], "counters": {}})
                e["counters"].update(v["counters"])
    except Exception as e:      # noqa: BLE001 -- the counters are an add-on: never fail the benchmark over them
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return merged, f"live: 3 rocprofv3 --pmc passes of this command's workloads ({time.time() - t0:.0f} s)"


def pmc_for_run(args, argv, live=True):
    """-> (table or None, source string)"""
    sha = library_sha256()
    if not live:
        note = "N > 1: no live passes"
    elif args.pmc == "auto":
        tab, note = collect_pmc_live(argv)
        if tab is not None:
            if args.pmc_dump:
                with open(args.pmc_dump, "w") as f:
                    json.dump({"library_sha256": sha, "steps": args.steps, "warmup": args.warmup, "workload": workload_key(args), "source": note,
                               "units": "mean per timed launch; FETCH_SIZE / WRITE_SIZE in KB (uncorrected)", "rows": tab}, f, indent=1)
            return tab, note
    else:
        note = "--pmc off"
    try:
        committed = json.load(open(PMC_TABLE))
    except (OSError, ValueError):
        return None, f"{note}; no committed table"
    if committed.get("library_sha256") != sha:
        return None, f"{note}; the committed table was measured on another library build"
    if committed.get("workload") != workload_key(args):
        return None, f"{note}; the committed table was measured on other workload arguments"
    return committed["rows"], (f"{note}; committed table profiles/r02_pmc_rows.json (same library sha256, same workload arguments, "
                               f"measured with --steps {committed.get('steps')})")


def traffic_bytes(c):
    """HBM bytes per launch: FETCH_SIZE x 2 (gfx950: 128-B read requests tallied at 64 B -- MI355X_MICROARCH.md for 16-B-per-lane
    loads, profiles/r01f_hbm_counter_calibration.txt for 4/8-B-per-lane loads) + WRITE_SIZE (exact), both reported in KB."""
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)


def exec_block(c, kernel_s):
    """Executed fp64 rate and VALU issue share from the counters of one launch."""
    if not c or "SQ_INSTS_VALU_FMA_F64" not in c:
        return None
    flops = 64.0 * (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"])
    slots = N_SIMD * kernel_s * NOMINAL_CLOCK_HZ / 4.0
    return {"tflops": flops / kernel_s / 1e12, "flops_per_launch": flops,
            "valu_issue_frac": c["SQ_INSTS_VALU"] / slots if "SQ_INSTS_VALU" in c else None,
            "fp64_arith_share_of_valu": (c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]) / c["SQ_INSTS_VALU"]
            if c.get("SQ_INSTS_VALU") else None}


def roofline_step(meas, n_local, solver, mixed, pmc_row, pmc_src):
    """fp64-VALU-bound rows (a full env-step)."""
    c = (pmc_row or {}).get("counters")
    t = meas["kernel_ms_avg"] * 1e-3
    ex = exec_block(c, t)
    flops_per_unit = FLOPS_PER_RK45_ATTEMPT if solver == "rk45" else FLOPS_PER_RK4_SUBSTEP
    units_per_launch = meas["work_units"] / meas["launches"]
    we = flops_per_unit * units_per_launch / t / 1e12
    bytes_per_launch = (BYTES_PER_ENV_STEP_MIXED if mixed else BYTES_PER_ENV_STEP) * n_local
    gbs = bytes_per_launch / t / 1e9
    return {"bound": "valu_fp64", "achieved": round(ex["tflops"], 4) if ex else None, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ex["tflops"] / PEAK_FP64_VALU_TFLOPS, 5) if ex else None,
            "traffic": traffic_bytes(c), "algorithmic_bytes": bytes_per_launch,
            "basis": "executed: 64 x (2 FMA_F64 + MUL_F64 + ADD_F64) wavefront instructions per launch (hardware counters) / "
                     "HIP-event kernel time",
            "valu_issue_frac": round(ex["valu_issue_frac"], 4) if ex and ex["valu_issue_frac"] is not None else None,
            "executed_flops_per_work_unit": round(ex["flops_per_launch"] / units_per_launch, 1) if ex else None,
            "kernel": (pmc_row or {}).get("kernel_name", "stg_step_kernel"), "kernel_ms_avg": round(meas["kernel_ms_avg"], 4),
            "work_units_per_env_step": round(meas["work_units"] / max(meas["env_steps"], 1), 2),
            "noop_frac": round(meas["noop_steps"] / max(meas["env_steps"], 1), 6),
            "work_equiv": {"tflops": round(we, 3), "flops_per_work_unit": flops_per_unit,
                           "note": "reference formulation's flop count (SURVEY 8d) x work units / time: how fast the reference's "
                                   "arithmetic gets done; the kernels fold constants and execute fewer -- NOT a roofline fraction"},
            "hbm": {"achieved": round(gbs, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 7),
                    "bytes_per_env_step": BYTES_PER_ENV_STEP_MIXED if mixed else BYTES_PER_ENV_STEP},
            "pmc_source": pmc_src}


def roofline_hbm(meas, bytes_per_launch, pmc_row, pmc_src, extra=None):
    """HBM-shaped rows (cfg2a, the array env): algorithmic bytes per launch over the kernel time."""
    c = (pmc_row or {}).get("counters")
    t = meas["kernel_ms_avg"] * 1e-3
    gbs = bytes_per_launch / t / 1e9
    ex = exec_block(c, t)
    out = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
           "traffic": traffic_bytes(c), "algorithmic_bytes": bytes_per_launch,
           "kernel": (pmc_row or {}).get("kernel_name", meas.get("kernel", "stg_step_kernel")), "kernel_ms_avg": round(meas["kernel_ms_avg"], 4),
           "valu_issue_frac": round(ex["valu_issue_frac"], 4) if ex and ex["valu_issue_frac"] is not None else None,
           "pmc_source": pmc_src}
    if ex and ex["valu_issue_frac"] is not None:
        # HBM-shaped by its bytes per unit, but say which resource the launch actually sits closer to
        out["closer_to"] = "valu_issue" if ex["valu_issue_frac"] > out["frac"] else "hbm"
    out.update(extra or {})
    return out


def usable_cores(omp_max):
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota if there is one."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(solver, thermal, seconds):
    """The oracle (CPU restatement, "port") on this box's host cores, same workload distribution, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from helpers import make_states, unit_rows
    import spin_torque_gym_amd as stg  # noqa: F401
    oracle.build()
    n = 2048
    rng = np.random.default_rng(3)
    m0 = unit_rows(rng, n)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    p = (oracle.Params * 1)(oracle.make_params(stt_params(volume_for(solver))))
    c = oracle.make_config(solver=solver, thermal=bool(thermal), seed=1234)
    threads = usable_cores(oracle.lib().stgo_max_threads())
    done_steps, t_used, batches = 0, 0.0, 0
    while t_used < seconds and batches < 4096:
        st = make_states(n, m0, tgt)
        a = np.empty((n, 2), dtype=np.float32)
        a[:, 0] = rng.uniform(-2e6, 2e6, n)
        a[:, 1] = rng.uniform(1e-10, 1e-9, n)
        t0 = time.perf_counter()
        oracle.env_step_batch(st, a, p, None, c, env_id0=0, n_threads=threads)
        t_used += time.perf_counter() - t0
        done_steps += n
        batches += 1
    return {"value": round(done_steps / t_used, 1), "unit": "env-steps/s", "cores": int(threads), "kind": "port",
            "sample": f"{batches} batches x {n} envs, one env.step each, same action distribution, solver={solver}, "
                      f"thermal={int(bool(thermal))}, OpenMP over envs ({t_used:.1f} s of wall time)"}


def child_argv(args):
    return ["--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--envs-per-gpu", str(args.envs_per_gpu),
            "--solver", args.solver, "--thermal", str(args.thermal), "--also", str(args.also), "--lane-sort", args.lane_sort,
            "--wave-spec", args.wave_spec, "--pmc", "off"]


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.pmc_child:
        pmc_child(args)
        return
    if os.environ.get("STG_BENCH_DEBUG"):
        import gc
        _g0 = [0.0]

        def _gc_all(phase, info):        # every collector pass of this process, with its duration
            if phase == "start":
                _g0[0] = time.perf_counter()
            elif info["generation"] >= 1:
                print("debug gc: generation %d pass took %.3f ms (collected %d) at t=%.3f s" % (
                    info["generation"], (time.perf_counter() - _g0[0]) * 1e3, info["collected"], time.perf_counter()), file=sys.stderr, flush=True)
        gc.callbacks.append(_gc_all)
    # hardware counters first: the child passes must be started before this process initialises the GPU
    pmc_tab, pmc_src = pmc_for_run(args, child_argv(args), live=(world == 1)) if rank == 0 else (None, "rank > 0")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(ndev, 1)          # rehearsal: several ranks may share one GPU
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)
    n_local = args.envs_per_gpu
    specs = row_specs(args)
    row = lambda key: (pmc_tab or {}).get(key)
    meas = run_row(specs[0][1], rank, world, local_rank, gather_algo=args.gather_algo)
    wall = torch.tensor([meas["wall_s"]], dtype=torch.float64, device=torch.device("cuda", local_rank))
    if world > 1:
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    wall_s = float(wall.item())
    n_total = n_local * world
    out = {
        "metric": "env_steps_per_sec", "value": round(n_total * args.steps / wall_s, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall_s / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "blocks_timed": meas["blocks_timed"],
        "no_gather": None,
        "config": {"workload": f"cfg3: {n_local} STT-MRAM envs/GPU, thermal {'on 300K (in-kernel Philox)' if args.thermal else 'off'}, "
                               f"solver={args.solver} ({'LLGSSolver SciPy-RK45 rtol1e-6 atol1e-9 max_step 1ps' if args.solver == 'rk45' else 'SimpleLLGSSolver fixed-step dt<=1ps'}), "
                               f"full env.step, J~U[-2e6,2e6], pulse~U[0.1,1]ns f32, volume={volume_for(args.solver):g}, autoreset",
                   "envs_per_gpu": n_local, "global_envs": n_total, "solver": args.solver, "thermal": bool(args.thermal),
                   "parallelism": (f"env-sharded x{world}; per step ONE {args.gather_algo} of 56 B/env records (written by the step kernel straight into the send buffer) into the "
                                   f"global record array, pipelined under the next kernel; the timed loop hands out the learner's typed views "
                                   f"(obs [N,12], reward, terminated, truncated) -- no copies") if world > 1 else "single GPU"},
        "roofline": roofline_step(meas, n_local, args.solver, False, row("headline"), pmc_src),
    }
    if world > 1:
        wng = torch.tensor([meas["wall_no_gather_s"]], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(wng, op=dist.ReduceOp.MAX)
        out["no_gather"] = {"value": round(n_total * args.steps / float(wng.item()), 1), "unit": "env-steps/s",
                            "ms_per_step": round(float(wng.item()) / args.steps * 1e3, 4),
                            "note": "same run without the per-step all-gather (learner data-parallel over the same ranks)"}
        gm = torch.tensor([meas["gather_only_ms"]], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)
        out["gather_only"] = {"ms": round(float(gm.item()), 4), "bytes_per_rank": 56 * n_local, "algo": args.gather_algo,
                              "note": "the exchange by itself, back to back with nothing to hide under (max over ranks); in the "
                                      "timed loop it runs on its own stream under the next step's kernel"}
    if rank == 0 and world == 1 and args.also:
        also = []
        for key, spec in specs[1:]:
            m = run_row(spec, 0, 1, local_rank)
            st = spec["steps"]
            if spec["kind"] == "step":
                also.append({"workload": key, "value": round(spec["n"] * st / m["wall_s"], 1), "unit": "env-steps/s",
                             "ms_per_step": round(m["wall_s"] / st * 1e3, 4),
                             "roofline": roofline_step(m, spec["n"], spec["solver"], spec["mixed"], row(key), pmc_src)})
            elif spec["kind"] == "short":
                also.append({"workload": m["workload"], "value": round(m["n"] * m["K"] * st / m["wall_s"], 1), "unit": "env-steps/s",
                             "ms_per_step": round(m["wall_s"] / st / m["K"] * 1e3, 5),
                             "roofline": roofline_hbm(m, m["bytes_per_launch"], row(key), pmc_src,
                                                      {"work_units_per_env_step": round(m["work_units"] / max(m["env_steps"], 1), 2),
                                                       "noop_frac": round(m["noop_steps"] / max(m["env_steps"], 1), 6)})})
            else:
                also.append({"workload": m["workload"], "value": round(m["n"] * st / m["wall_s"], 1), "unit": "array-steps/s",
                             "ms_per_step": round(m["wall_s"] / st * 1e3, 4),
                             "roofline": roofline_hbm(m, m["bytes_per_unit"] * m["n"], row(key), pmc_src,
                                                      {"bytes_per_array_step": m["bytes_per_unit"]})})
        out["also"] = also
    if rank == 0 and world == 1 and args.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.solver, args.thermal, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
