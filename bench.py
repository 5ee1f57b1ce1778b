#!/usr/bin/env python3
"""bench.py -- env-steps/s of the SpinTorque-v0 step path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU -- launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                (the same ranks under an external launcher)

A "step" is one env.step() of every environment of the batch: one launch of the fused step kernel per GPU (action
clamp -> LLGS integration over the pulse -> energy -> observation -> reward -> termination), and for N > 1 the single
all-gather of (obs, reward, terminated, truncated).  Inputs (states, the [K,2,N] action tensor) are resident in HBM
before the timed region.  Weak scaling: --envs-per-gpu environments per GPU whatever N is.

Headline workload (config.workload): BASELINE.json configs[2] / the north-star target -- 65 536 STT-MRAM environments
per GPU, thermal field on at 300 K (in-kernel Philox), LLGSSolver semantics (SciPy RK45, rtol 1e-6, atol 1e-9,
max_step 1 ps), random pulses J ~ U[-2e6, 2e6] A/m^2, duration ~ U[0.1, 1] ns (float32), `volume` rescaled to 9.7e-6 so
that the Slonczewski term is well conditioned for RK45 (SURVEY.md headline 3: at the default volume any J != 0 makes
the reference's solver diverge), device-side auto-reset of finished episodes, the env's default outputs (diagnostics
off: obs / reward / terminated / truncated records only).  `also` carries the other configurations (cfg2: 4096 envs
T = 0 K; the env's own RK4 solver; one cfg5 shard; cfg4: 262 144 mixed STT/SOT/VCMA envs with a class table, with the
device-physics torque terms, and with one parameter record per env; cfg2a; the array env).  Every configuration timed
here is compared with the oracle at the same size in tests/test_gpu_fullsize.py.

What is timed.  `value` / `ms_per_step`: host wall time of a block of exactly --steps calls of the C-ABI step
(HipBackend.step -> stg_step_many) between synchronizes (+ barriers for N > 1), max over ranks; --blocks (3) such blocks
are timed and the MEDIAN one is reported -- where the dispatcher places the wavefronts differs from launch to launch by a
few per cent, one block cannot tell a regression from placement luck.  `api_ms_per_step`: the
same steps through the public SpinTorqueVecEnv.step() with Gym-convention [N, 2] actions (headline and cfg2).
Timing hygiene (profiles/r03_stall_root_cause.txt): the GPU boxes show 256 CPUs but the container's CFS quota is 16; torch's
default pool of 128 host threads made the container get THROTTLED for up to 100 ms right after a row's set-up -- the
"sporadic ~80 ms stall" of rounds 1-2.  The pool is capped (cap_host_threads), the collector is off inside a block, events
exist before the block, and a block is discarded and replaced (at most twice) ONLY if cgroup cpu.stat reports a throttling
event during it -- never on the measured times.  Every block's wall time is printed: `block_walls_ms`, `block_throttled`,
`block_used`, `blocks_timed` are the LAST keys of the line.

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
usable, the committed table profiles/r04_pmc_rows.json is used when it was measured on this very library build (sha256
of the .so), and `frac`/`traffic` are null otherwise: no stale number is ever printed.

N > 1: rank 0 runs the counter passes for the headline row (a rank's step kernel is the single-GPU kernel of its shard) before
it joins the process group, and the CPU baseline after the timed region while the other ranks wait -- the line carries
`roofline` and `cpu_baseline` like the N = 1 line.  check_ranks() asserts that the process group spans --gpus ranks on as many
distinct GPUs (RCCL); besides the pipelined loop the line carries `no_gather` (no exchange at all) and `gather_only` with the
exchange by itself for all three forms (all_gather, in-place all_gather, one-shot p2p) in one run; if that sweep hangs, the
measured line is still printed (with `error`) and every rank exits with code 3.
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
BYTES_PER_ENV_STEP_PER_ENV = 272  # + 112 B: 14 fp64 parameters per env ("when params are per-env"); the library's
                                  # record is the whole stg_device_params (30 fp64 + 2 B = 242 B), see `traffic`
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
PMC_TABLE = os.path.join(ROOT, "profiles", "r04_pmc_rows.json")
LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
              "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
# (prefixes: every env-step / array-step kernel of the library, whatever its variant is called -- a kernel missing from an explicit
# list silently cost the whole counter table twice)
MAIN_KERNELS = ("stg_step_", "stg_array_step_")
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
    ap.add_argument("--sweep-timeout", type=float, default=120.0,
                    help="N > 1: seconds the exchange sweep (all_gather / in-place / p2p by themselves) may take before it is abandoned")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: collect the hardware counters live with rocprofv3 child passes (N = 1 only)")
    ap.add_argument("--pmc-dump", default=None, help="write the per-row counter table (JSON) here as well")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)     # internal: run the rows once, write a manifest
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps per row; the MEDIAN block is reported")
    ap.add_argument("--master-port", type=int, default=0, help="--gpus N > 1 without a launcher: rendezvous port of the ranks this "
                    "command starts itself (0 = pick a free one)")
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


def host_limits():
    """(CPUs the scheduler shows, CPUs this container may actually use per its CFS bandwidth quota)."""
    vis = os.cpu_count() or 1
    return vis, usable_cores(vis)


def cap_host_threads():
    """The GPU boxes show 256 CPUs but the container's CFS bandwidth quota is 16 (cgroup cpu.max = 1600000/100000): torch's
    default intra-op pool of 128 OpenMP workers wakes up for every elementwise op / copy of the input generation, spins after
    the parallel region and burns the quota of the 100 ms period, the kernel then THROTTLES the whole container until the next
    period -- and a thread asleep in hipDeviceSynchronize cannot be woken meanwhile.  That was the "sporadic ~80 ms stall, once
    per process" of rounds 1-2 (profiles/r03_stall_*: every stalled block coincides with nr_throttled += 1 in cpu.stat; none
    without).  The pool is capped well below the quota; the step path itself is single-threaded on the host."""
    vis, quota = host_limits()
    n = max(1, min(4, quota))
    torch.set_num_threads(n)
    return {"cpus_visible": vis, "cpu_quota": quota, "torch_threads": n}


class BlockTimer:
    """Times blocks of EXACTLY `steps` calls of `body(k)` between synchronize (+ barrier) on both sides.

    * every event exists and has been recorded once before the first block (torch creates the HIP event at the first
      record()): nothing is created or allocated by the timing itself inside a block;
    * the Python collector is off inside a block (one full pass before it): a generation-2 pass of this process takes
      35-95 ms on the GPU boxes;
    * the container's CFS throttling counters (cgroup cpu.stat) are read on both sides: `throttled` > 0 means the kernel
      suspended this container during the block -- the measurement then contains up to 100 ms that are neither the GPU's nor
      the program's, and the caller may time another block.  That is the only re-timing criterion; it does not look at
      the times measured."""

    def __init__(self, dev, steps, world=1):
        self.dev, self.steps, self.world = dev, steps, world
        self.starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        self.ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        self.fin = torch.cuda.Event(enable_timing=True)
        for e in (*self.starts, *self.ends, self.fin):
            e.record()
        torch.cuda.synchronize(dev)

    def block(self, body, tail=None):
        import gc
        import torch.distributed as dist
        gc.collect()
        gc.disable()
        try:
            cg0 = cgroup_cpu_stat()
            if self.world > 1:
                dist.barrier()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for k in range(self.steps):
                self.starts[k].record()
                body(k)
                self.ends[k].record()
            if tail is not None:
                tail()
            self.fin.record()
            torch.cuda.synchronize(self.dev)
            if self.world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            cg1 = cgroup_cpu_stat()
        finally:
            gc.enable()
        return {"wall_s": t1 - t0, "kernel_ms": [s_.elapsed_time(e) for s_, e in zip(self.starts, self.ends)],
                "device_span_s": self.starts[0].elapsed_time(self.fin) * 1e-3,
                "throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                "throttled_usec": cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)}

    def timed(self, body, tail=None, retime=True, tag="", pre=None, post=None, nblocks=None):
        """-> (the MEDIAN block, all blocks).  `nblocks` blocks (default --blocks = 3) of exactly `steps` steps each are timed;
        a block during which the container was throttled is DISCARDED and replaced (twice at most; if every block was throttled
        all are kept) -- the throttle counter decides what is discarded, the measured times never decide what is selected: the
        reported block is the median of the kept ones by wall time (max over ranks).  `pre` runs before every block and `post(b)`
        after it, both outside the timed region (resetting / reading the on-device work counters).  retime=False: one block."""
        import torch.distributed as dist
        want = 1 if not retime else max(1, int(nblocks if nblocks is not None else DEFAULT_BLOCKS))
        blocks = []
        while True:
            if pre is not None:
                pre()
            b = self.block(body, tail)
            v = [float(b["throttled"] > 0), b["wall_s"]]
            if self.world > 1:
                t = torch.tensor(v, dtype=torch.float64, device=self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                v = t.tolist()
            b["discard"], b["wall_max_s"] = bool(v[0]), float(v[1])
            if post is not None:
                post(b)
            blocks.append(b)
            if os.environ.get("STG_BENCH_DEBUG"):
                print("debug %s: block %d wall %.3f ms, device span %.3f ms, sum of kernel times %.3f ms, throttled %d (%d us)" % (
                    tag, len(blocks), b["wall_s"] * 1e3, b["device_span_s"] * 1e3, sum(b["kernel_ms"]), b["throttled"], b["throttled_usec"]),
                    file=sys.stderr, flush=True)
            clean = [x for x in blocks if not x["discard"]]
            if len(clean) >= want or len(blocks) >= want + 2:
                break
        kept = clean if clean else blocks
        med = sorted(kept, key=lambda x: x["wall_max_s"])[(len(kept) - 1) // 2]
        for x in blocks:
            x["used"] = x is med
        return med, blocks


DEFAULT_BLOCKS = 3


def block_report(blocks):
    """every block's wall time (max over ranks), which were discarded as throttled, which one (the median of the rest) is reported"""
    return {"blocks_timed": len(blocks), "block_walls_ms": [round(b["wall_max_s"] * 1e3, 4) for b in blocks],
            "block_throttled": [int(b["throttled"]) for b in blocks], "block_used": [i for i, b in enumerate(blocks) if b["used"]][0]}


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


def per_env_variation(n, seed=11):
    """Device-to-device variation of the per-env-parameter cfg4 row (and of its test, tests/test_gpu_fullsize.py): every
    env gets its own volume, damping, anisotropy, magnetisation, polarisation and parallel resistance around its base
    device's values (reference keys; arrays of length n)."""
    rng = np.random.default_rng(seed)
    return {"volume_scale": 10 ** rng.uniform(-0.15, 0.15, n), "damping": 10 ** rng.uniform(-2.2, -1.3, n),
            "uniaxial_anisotropy": rng.uniform(8e5, 1.2e6, n), "saturation_magnetization": rng.uniform(7e5, 9e5, n),
            "polarization": rng.uniform(0.5, 0.9, n), "resistance_parallel": rng.uniform(900.0, 1300.0, n)}


def mixed_kwargs(solver, n_global, per_env=False):
    """cfg4: class = env index mod 3 over STT / SOT / VCMA, factory defaults of each type with polarization 0.7 and the
    rescaled volume; per_env: on top of that every env's own record (stg_set_params_per_env)."""
    import spin_torque_gym_amd as stg
    fac = stg.DeviceFactory()
    sot = fac.get_default_parameters("sot_mram"); sot.update(polarization=0.7, volume=volume_for(solver))
    vc = fac.get_default_parameters("vcma_mram"); vc.update(polarization=0.7, volume=volume_for(solver))
    kw = dict(device_type=["stt_mram", "sot_mram", "vcma_mram"], device_params=[stt_params(volume_for(solver)), sot, vc])
    cls = (torch.arange(n_global) % 3).to(torch.uint8)
    if per_env:
        v = per_env_variation(n_global)
        ov = {k: val for k, val in v.items() if k != "volume_scale"}
        ov["volume"] = volume_for(solver) * v["volume_scale"]
        kw["per_env_params"] = ov
    return kw, cls


def run_config(n_local, solver, thermal, steps, warmup, rank, world, device_index, mixed=False, seed=1234, lane_sort=None,
               torque_model="reference", wave_spec=None, gather_algo="all_gather", retime=True, per_env=False, api=False,
               inplace=False, extras=True):
    """Builds the env, runs warmup + timed steps, returns a dict of measurements (times are this rank's).
    The env is the product default: diagnostics off (a step writes the RL-facing outputs only), records layout."""
    import spin_torque_gym_amd as stg
    import torch.distributed as dist
    kw = dict(include_thermal_fluctuations=bool(thermal), temperature=300.0, solver=solver, seed=seed, autoreset=True,
              lane_sort=lane_sort, torque_model=torque_model, wave_spec=wave_spec)
    if mixed:
        mk, cls_global = mixed_kwargs(solver, n_local * world, per_env)
        kw.update(mk)
    else:
        kw.update(device_params=stt_params(volume_for(solver)))
        cls_global = None
    if world > 1:
        from spin_torque_gym_amd.distributed import ShardedSpinTorqueVecEnv
        env = ShardedSpinTorqueVecEnv(n_local * world, device_index=device_index, class_index=cls_global,
                                      gather_algo=gather_algo, inplace=inplace, **kw)
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
        k-1's all-gather (own stream) runs under step k's kernel; gather_end hands out typed views -- what a learner
        consumes, no copies (spin_torque_gym_amd/distributed.py)."""
        if world == 1:
            backend.step(acts[k], autoreset=True)                    # the step kernel, on torch's current stream
            return
        env.step(acts[k], gather=False, actions_are_local=True, actions_soa=True)
        if gather:
            if env.gather_in_flight:
                env.gather_end()                                     # step k-1's gather ran under step k's kernel
            env.gather_begin()                                       # the single collective of a step, on its own stream

    def drain():
        if world > 1 and env.gather_in_flight:
            env.gather_end()                                         # (typed global views: obs [N,12], reward, flags)

    for k in range(warmup):
        one_step(k)
    drain()
    torch.cuda.synchronize(dev)
    timer = BlockTimer(dev, steps, world)
    tag = "n=%d %s th=%s tm=%s%s" % (n_local, solver, thermal, torque_model, " per-env" if per_env else "")
    def keep_counters(b):
        b["counters"] = backend.counters()                           # work units of THIS block (reset before it)
        # ... and where the dispatcher put the wavefronts of each of its launches (stg_get_placement keeps the last 32 launches)
        b["placement"] = [backend.placement(steps - 1 - k) for k in range(steps)] if steps <= 32 else None

    last, blocks = timer.timed(lambda k: one_step(warmup + k), drain, retime, tag, pre=lambda: backend.counters(reset=True),
                               post=keep_counters)
    c = last["counters"]                                             # ... of the reported (median) block
    out = dict(wall_s=last["wall_max_s"], placement=last["placement"], kernel_ms=[round(x, 4) for x in last["kernel_ms"]], device_span_s=last["device_span_s"], kernel_ms_avg=float(np.mean(last["kernel_ms"])),
               kernel_ms_min=float(np.min(last["kernel_ms"])), env_steps=c["env_steps"], work_units=c["work_units"],
               noop_steps=c["noop_steps"], launches=steps, warmup=warmup, **block_report(blocks))
    out.update(gather_only_ms=None, wall_no_gather_s=None, api_ms_per_step=None, lane_sort=lane_sort)
    if world > 1 and extras:
        # SURVEY 8e "report both": a learner that is data-parallel over the same ranks needs no gather at all
        out["wall_no_gather_s"] = timer.timed(lambda k: one_step(warmup + k, gather=False), None, retime, tag + " no-gather")[0]["wall_max_s"]
    if world > 1:
        out["gather_only_ms"] = gather_only(env, acts[warmup], steps, dev)
    if api and world == 1:
        # the public API: SpinTorqueVecEnv.step() with Gym-convention [N, 2] actions (host-side views, the transposition of
        # the actions into the kernel's [2, N], the timer table, the info dict) -- what a user's loop pays per step
        acts_gym = acts[warmup:].transpose(1, 2).contiguous()
        env.step(acts_gym[0])
        torch.cuda.synchronize(dev)
        out["api_ms_per_step"] = timer.timed(lambda k: env.step(acts_gym[k]), None, retime, tag + " api")[0]["wall_max_s"] / steps * 1e3
    env.close()
    return out


def gather_only(env, act, steps, dev):
    """The exchange by itself (nothing to hide under): `steps` exchanges of one step's records back to back, ms each."""
    import torch.distributed as dist
    env.step(act, gather=False, actions_are_local=True, actions_soa=True)
    env.gather_begin()
    env.gather_end()                                                 # (first use of this algorithm's communicator path)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.gather_again()                                           # (re-send the records of the step above)
        env.gather_begin()
        env.gather_end()
    torch.cuda.synchronize(dev)
    dist.barrier()
    return (time.perf_counter() - t0) / steps * 1e3


def gather_only_all_algos(n_local, solver, thermal, steps, rank, world, device_index, seed=1234):
    """N > 1: the exchange by itself for the three forms -- RCCL all-gather out of place (default), in place, and the
    one-shot point-to-point exchange -- in ONE run, so that the first multi-GPU lease yields the choice."""
    import torch.distributed as dist
    from spin_torque_gym_amd.distributed import ShardedSpinTorqueVecEnv
    out = {}
    for name, algo, inplace in (("all_gather", "all_gather", False), ("all_gather_inplace", "all_gather", True), ("p2p", "p2p", False)):
        try:
            env = ShardedSpinTorqueVecEnv(n_local * world, device_index=device_index, gather_algo=algo, inplace=inplace,
                                          include_thermal_fluctuations=bool(thermal), solver=solver, seed=seed, autoreset=True,
                                          device_params=stt_params(volume_for(solver)))
            dev = env.local.backend.device
            env.reset(seed=seed + rank)
            act = make_actions(1, n_local, dev, seed + 17 * rank)[0]
            ms = torch.tensor([gather_only(env, act, steps, dev)], dtype=torch.float64, device=dev)
            dist.all_reduce(ms, op=dist.ReduceOp.MAX)
            out[name] = round(float(ms.item()), 4)
            env.close()
        except Exception as e:          # noqa: BLE001 -- an algorithm RCCL refuses must not cost the run its headline number
            out[name] = f"failed: {type(e).__name__}: {e}"[:200]
    return out


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
    last, blocks = BlockTimer(dev, steps).timed(lambda k: env.backend.step(acts[k + 2]), None, retime, f"array {mode}")
    env.close()
    affected = {"individual": 1, "row": size[1], "column": size[0], "global": ndev}[mode]
    # algorithmic bytes per array-step: pattern + target + state + action read; addressed cells, obs, reward, flags, state written
    b = (ndev * 24 * 2 + 12 + 4 * a_dim) + (affected * 24 + ndev * 24 + 4 + 2 + 12 + 16)
    return dict(kind="array", wall_s=last["wall_max_s"], kernel_ms_avg=float(np.mean(last["kernel_ms"])), launches=steps, warmup=2, n=n, bytes_per_unit=b,
                **block_report(blocks),
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
    call = (lambda k: b.step(a[0], autoreset=True)) if K == 1 else (lambda k: b.step_many(a, out_every=False, autoreset=True))
    for _ in range(2):
        call(0)
    torch.cuda.synchronize(b.device)
    def keep_counters(blk):
        blk["counters"] = b.counters()

    last, blocks = BlockTimer(b.device, steps).timed(call, None, retime, f"cfg2a K={K}", pre=lambda: b.counters(reset=True), post=keep_counters)
    c = last["counters"]
    env.close()
    # state read and written once per launch, actions read K times, outputs written once (out_every = False)
    return dict(kind="short", wall_s=last["wall_max_s"], kernel_ms_avg=float(np.mean(last["kernel_ms"])), launches=steps, warmup=2, n=n, K=K,
                bytes_per_launch=BYTES_PER_ENV_STEP * n + 8 * n * (K - 1), env_steps=c["env_steps"], work_units=c["work_units"],
                noop_steps=c["noop_steps"], **block_report(blocks),
                workload=f"cfg2a: {n} STT envs, T=0K, rk45, every pulse = one 1 ps DP5 step, {K} env-step(s) per launch")


# ----------------------------------------------------------------------------------------------------------------------
# the rows: headline first, then the secondary configurations (N = 1 only)
# ----------------------------------------------------------------------------------------------------------------------
def row_specs(args):
    """(key, runner) in execution order -- the same list drives the timed parent run and the PMC child passes."""
    lane_sort = {"auto": None, "on": True, "off": False}[args.lane_sort]
    wave_spec = {"auto": None, "on": True, "off": False}[args.wave_spec]
    st = max(3, args.steps // 2)
    rows = [("headline", dict(kind="step", n=args.envs_per_gpu, solver=args.solver, thermal=args.thermal, mixed=False, per_env=False,
                              tm="reference", steps=args.steps, warmup=args.warmup, lane_sort=lane_sort, wave_spec=wave_spec, api=True))]
    if not args.also:
        return rows
    for name, n, solver, thermal, mixed, tm, per_env, api in (
            ("cfg2: 4096 STT envs, T=0K, rk45", 4096, "rk45", 0, False, "reference", False, True),
            ("cfg2: 4096 STT envs, T=0K, rk4 (the env's own solver)", 4096, "rk4", 0, False, "reference", False, False),
            ("cfg3: 65536 STT envs, thermal on, rk4", 65536, "rk4", 1, False, "reference", False, False),
            ("cfg5 shard: 131072 STT envs (1 048 576 over 8 GPUs), thermal on, rk45", 131072, "rk45", 1, False, "reference", False, False),
            ("cfg5 on ONE GPU: 1048576 STT envs, thermal on, rk45 (lane-refill kernel: 8 envs per lane)", 1048576, "rk45", 1, False, "reference", False, False),
            ("cfg4: 262144 mixed STT/SOT/VCMA envs (class table in LDS), T=0K, rk4, reference RHS for all types",
             262144, "rk4", 0, True, "reference", False, False),
            ("cfg4: 262144 mixed STT/SOT/VCMA envs, T=0K, rk4, device-physics torque terms per type (opt-in)",
             262144, "rk4", 0, True, "device", False, False),
            ("cfg4 per-env: 262144 mixed STT/SOT/VCMA envs, every env its own parameter record (stg_set_params_per_env), T=0K, rk4",
             262144, "rk4", 0, True, "reference", True, False)):
        if solver == args.solver and n == args.envs_per_gpu and bool(thermal) == bool(args.thermal) and not mixed:
            continue
        rows.append((name, dict(kind="step", n=n, solver=solver, thermal=thermal, mixed=mixed, per_env=per_env, tm=tm, steps=st, warmup=2,
                                lane_sort=None, wave_spec=None, api=api)))
    for K in (1, 8):
        rows.append((f"cfg2a K={K}", dict(kind="short", n=1048576, K=K, steps=st)))
    for mode in ("individual", "global"):
        rows.append((f"array {mode}", dict(kind="array", n=262144, mode=mode, steps=st)))
    return rows


def run_row(spec, rank, world, local_rank, retime=True, gather_algo="all_gather", api=True):
    if spec["kind"] == "step":
        m = run_config(spec["n"], spec["solver"], spec["thermal"], spec["steps"], spec["warmup"], rank, world, local_rank,
                       mixed=spec["mixed"], torque_model=spec["tm"], lane_sort=spec["lane_sort"], wave_spec=spec["wave_spec"],
                       gather_algo=gather_algo, retime=retime, per_env=spec["per_env"], api=api and spec["api"])
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
        m = run_row(spec, 0, 1, 0, retime=False, api=False)
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


def workload_key(args, world=1):
    """What the rows of a counter table depend on besides the library build."""
    return {"envs_per_gpu": args.envs_per_gpu, "solver": args.solver, "thermal": int(bool(args.thermal)),
            "also": int(bool(args.also)) if world == 1 else 0, "lane_sort": args.lane_sort, "wave_spec": args.wave_spec}


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
    # (a rank of a multi-process run starts these passes too: the single-GPU child must not see the launcher's rendezvous variables)
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV and not k.startswith("TORCHELASTIC_")}
    env["TMPDIR"] = "/tmp"
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


def pmc_for_run(args, argv, world=1):
    """-> (table or None, source string).  N > 1: rank 0 runs the passes for the headline row on its own GPU before it joins the
    process group (the other ranks wait at the rendezvous)."""
    sha = library_sha256()
    wk = workload_key(args, world)
    if args.pmc == "auto":
        tab, note = collect_pmc_live(argv)
        if tab is not None:
            if args.pmc_dump:
                with open(args.pmc_dump, "w") as f:
                    json.dump({"library_sha256": sha, "steps": args.steps, "warmup": args.warmup, "workload": wk, "source": note,
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
    cw = committed.get("workload")
    if cw != wk and not (world > 1 and cw == dict(wk, also=cw.get("also") if isinstance(cw, dict) else 0)):
        return None, f"{note}; the committed table was measured on other workload arguments"
    return committed["rows"], (f"{note}; committed table profiles/{os.path.basename(PMC_TABLE)} (same library sha256, same workload arguments, "
                               f"measured with --steps {committed.get('steps')})")


def traffic_bytes(c, coalesced_read_bytes=None):
    """HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (both reported in KB).  gfx950 tallies the 128-B read requests of a coalesced
    stream at 64 B: FETCH_SIZE reads HALF the bytes there (MI355X_MICROARCH.md for 16-B-per-lane loads;
    profiles/r01f_hbm_counter_calibration.txt for 4/8-B-per-lane loads) -- and EXACTLY the bytes of scattered 64-byte records, whether
    64, 8 or 1 lanes of a wavefront read one each (profiles/r04_hbm_scatter_calibration.txt: what a lane of the duration-sorted
    schedule or a refill point does with its env's state record).  WRITE_SIZE is exact (sector-granular: a scattered 56-byte record
    costs 80-85 B, same file).
    coalesced_read_bytes = None: every read of the launch is a coalesced stream (identity schedule, the array env): 2 x FETCH_SIZE.
    Otherwise the launch reads scattered records plus `coalesced_read_bytes` of coalesced streams (the step's actions and the
    permutation, in slot order): FETCH_SIZE = scattered + coalesced / 2, so reads = FETCH_SIZE + coalesced / 2 (never more than
    2 x FETCH_SIZE)."""
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
    reads = 2.0 * fetch if coalesced_read_bytes is None else min(2.0 * fetch, fetch + 0.5 * coalesced_read_bytes)
    return int(reads + write)


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


def roofline_step(meas, n_local, solver, mixed, pmc_row, pmc_src, per_env=False):
    """fp64-VALU-bound rows (a full env-step)."""
    bpe = BYTES_PER_ENV_STEP_PER_ENV if per_env else (BYTES_PER_ENV_STEP_MIXED if mixed else BYTES_PER_ENV_STEP)
    c = (pmc_row or {}).get("counters")
    t = meas["kernel_ms_avg"] * 1e-3
    ex = exec_block(c, t)
    flops_per_unit = FLOPS_PER_RK45_ATTEMPT if solver == "rk45" else FLOPS_PER_RK4_SUBSTEP
    units_per_launch = meas["work_units"] / meas["launches"]
    we = flops_per_unit * units_per_launch / t / 1e12
    bytes_per_launch = bpe * n_local
    gbs = bytes_per_launch / t / 1e9
    # under the duration-sorted schedule (every step row of this bench: random pulse durations) a lane's 64-B state record is a scattered
    # read, counted at face value; only the actions (8 B), the permutation (4 B) and the class index (1 B) are coalesced streams
    sorted_schedule = meas.get("lane_sort") is not False
    tb = traffic_bytes(c, (13 if mixed else 12) * n_local if sorted_schedule else None)
    tb_doubled = traffic_bytes(c)
    return {"bound": "valu_fp64", "achieved": round(ex["tflops"], 4) if ex else None, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ex["tflops"] / PEAK_FP64_VALU_TFLOPS, 5) if ex else None,
            "traffic": tb, "algorithmic_bytes": bytes_per_launch,
            "traffic_over_algorithmic": round(tb / bytes_per_launch, 3) if tb else None,
            "traffic_basis": ("FETCH_SIZE (scattered 64-B state records: counted exactly) + half the coalesced streams' bytes (actions, "
                              "permutation: counted at half) + WRITE_SIZE; calibration profiles/r04_hbm_scatter_calibration.txt"
                              if sorted_schedule else "2 x FETCH_SIZE (coalesced streams) + WRITE_SIZE"),
            "traffic_if_all_reads_doubled": tb_doubled,
            "basis": "executed: 64 x (2 FMA_F64 + MUL_F64 + ADD_F64) wavefront instructions per launch (hardware counters) / "
                     "HIP-event kernel time",
            "valu_issue_frac": round(ex["valu_issue_frac"], 4) if ex and ex["valu_issue_frac"] is not None else None,
            "executed_flops_per_work_unit": round(ex["flops_per_launch"] / units_per_launch, 1) if ex else None,
            "kernel": (pmc_row or {}).get("kernel_name", "stg_step_kernel"), "kernel_ms_avg": round(meas["kernel_ms_avg"], 4),
            "placement": placement_report(meas),
            "work_units_per_env_step": round(meas["work_units"] / max(meas["env_steps"], 1), 2),
            "noop_frac": round(meas["noop_steps"] / max(meas["env_steps"], 1), 6),
            "work_equiv": {"tflops": round(we, 3), "flops_per_work_unit": flops_per_unit,
                           "note": "reference formulation's flop count (SURVEY 8d) x work units / time: how fast the reference's "
                                   "arithmetic gets done; the kernels fold constants and execute fewer -- NOT a roofline fraction"},
            "hbm": {"achieved": round(gbs, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 7),
                    "bytes_per_env_step": bpe},
            "pmc_source": pmc_src}


def placement_report(meas):
    """What the dispatcher did with the launches of the reported block (the library records the SIMD of every wavefront of a launch:
    stg_get_placement).  simd_double_booked[k] = SIMDs that held two or more INTEGRATING wavefronts of launch k; the schedules of the
    launches of up to 65 536 envs (one integrating wavefront per SIMD; wave-specialised pairs) aim at 0, and a launch that got more
    is slow by placement, not by code: kernel_ms lists the same launches' HIP-event times."""
    pl = meas.get("placement")
    if not pl:
        return None
    db = [p["simd_double_booked"] for p in pl]
    mean_of = lambda key: round(float(np.mean([p[key] for p in pl if p.get(key) is not None])), 4) if any(p.get(key) is not None for p in pl) else None
    return {"workgroups": pl[-1]["workgroups"], "waves_per_workgroup": pl[-1]["waves_per_workgroup"], "simds_used": [p["simds_used"] for p in pl],
            "simd_double_booked": db, "simd_double_booked_mean": round(float(np.mean(db)), 2), "kernel_ms": meas.get("kernel_ms"),
            "integrating_wavefronts_per_simd_last_launch": pl[-1]["integrating_per_simd"],
            # from the wavefronts' start / retire times: mean share of the launch during which a SIMD held an integrating wavefront,
            # and the share of the launch left once 90 % of the SIMDs have retired their last one (the makespan's tail)
            "simd_busy_frac": mean_of("simd_busy_frac"), "last_simd_alone_frac": mean_of("last_simd_alone_frac"),
            "span_us": mean_of("span_us")}


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
    threads = host_limits()[1]        # every CPU the container's quota allows (the torch pool's cap does not apply here)
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


def child_argv(args, world=1):
    """arguments of the single-GPU counter passes: the same rows -- for N > 1 the headline row only (every rank's step kernel is the
    single-GPU kernel of its shard: same envs per GPU, same records layout)"""
    return ["--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--envs-per-gpu", str(args.envs_per_gpu),
            "--solver", args.solver, "--thermal", str(args.thermal), "--also", str(args.also if world == 1 else 0), "--lane-sort", args.lane_sort,
            "--wave-spec", args.wave_spec, "--pmc", "off"]


def check_ranks(args, rank, local_rank, world):
    """N > 1: every rank reports (rank, local device index, PCI domain/bus/device of its GPU) through the process group
    itself; asserts that the collective library really spans `--gpus` ranks and, with RCCL, that no two ranks share a GPU
    (by PCI address where torch exposes it, else by the local device index)."""
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    pr = torch.cuda.get_device_properties(local_rank)
    have_pci = all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    ident = (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id) if have_pci else (-1, -1, local_rank)
    mine = torch.tensor([rank, local_rank, *ident], dtype=torch.int64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    rows = [tuple(int(x) for x in t.tolist()) for t in allr]
    if dist.get_world_size() != args.gpus or sorted(r[0] for r in rows) != list(range(args.gpus)):
        raise SystemExit(f"process group has ranks {sorted(r[0] for r in rows)}, expected 0..{args.gpus - 1}")
    gpus = {r[2:] for r in rows}
    if args.backend == "nccl" and len(gpus) != world:
        raise SystemExit(f"{world} RCCL ranks on {len(gpus)} distinct GPUs: {rows}")
    return {"ranks_seen": len(rows), "distinct_gpus": len(gpus), "backend": dist.get_backend(),
            "gpus": [("%04x:%02x:%02x" % r[2:]) if r[2] >= 0 else f"cuda:{r[4]}" for r in sorted(rows)]}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: starts the N ranks as direct children of this process (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment -- what `python -m torch.distributed.run --nproc-per-node N`
    would set), relays rank 0's JSON line to stdout (everything else to stderr) and returns the worst exit code.  This process makes
    no GPU call at all (`torch.cuda.device_count()` only counts devices) and never replaces itself."""
    import socket
    import threading
    n = args.gpus
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} over RCCL needs {n} GPUs, this node shows {ndev} (to rehearse the multi-rank path on fewer GPUs: "
              f"--backend gloo)", file=sys.stderr)
        return 2
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))

    def relay(r, p):
        for line in p.stdout:
            is_line = r == 0 and line.startswith("{") and '"metric"' in line
            (sys.stdout if is_line else sys.stderr).write(line if is_line else f"[rank {r}] {line}")
            (sys.stdout if is_line else sys.stderr).flush()

    threads = [threading.Thread(target=relay, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    # a rank that dies leaves the others in a collective that never completes: once one has failed the rest get 60 s, then are
    # killed (by PID: these are this process's own children)
    failed_at = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 60:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for t in threads:
        t.join(timeout=5)
    codes = [p.wait() for p in procs]
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
    return min(worst, 255)


def main():
    args = parse_args()
    global DEFAULT_BLOCKS
    DEFAULT_BLOCKS = max(1, args.blocks)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process starts the N ranks itself and never touches the GPU
        sys.exit(launch_ranks(args))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (it launches its own "
                         f"ranks) or under `python -m torch.distributed.run --nproc-per-node {args.gpus}`")
    host = cap_host_threads()
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
        print("debug host:", host, "cgroup cpu.stat at start:", {k: v for k, v in cgroup_cpu_stat().items() if "thrott" in k or k == "nr_periods"},
              file=sys.stderr, flush=True)
    # hardware counters first: the child passes must be started before this process initialises the GPU
    pmc_tab, pmc_src = pmc_for_run(args, child_argv(args, world), world) if rank == 0 else (None, "rank > 0")
    import torch.distributed as dist
    ranks = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        from datetime import timedelta
        # (generous: ranks > 0 wait here while rank 0 runs its counter passes, up to 3 x 300 s)
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timedelta(minutes=30))
        else:
            local_rank = local_rank % max(ndev, 1)          # rehearsal: several ranks may share one GPU
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend, timeout=timedelta(minutes=30))
        ranks = check_ranks(args, rank, local_rank, world)
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
        "api_ms_per_step": round(meas["api_ms_per_step"], 4) if meas.get("api_ms_per_step") else None,
        "no_gather": None,
        "config": {"workload": f"cfg3: {n_local} STT-MRAM envs/GPU, thermal {'on 300K (in-kernel Philox)' if args.thermal else 'off'}, "
                               f"solver={args.solver} ({'LLGSSolver SciPy-RK45 rtol1e-6 atol1e-9 max_step 1ps' if args.solver == 'rk45' else 'SimpleLLGSSolver fixed-step dt<=1ps'}), "
                               f"full env.step, J~U[-2e6,2e6], pulse~U[0.1,1]ns f32, volume={volume_for(args.solver):g}, autoreset, "
                               f"diagnostics off (the step writes obs/reward/terminated/truncated only)",
                   "envs_per_gpu": n_local, "global_envs": n_total, "solver": args.solver, "thermal": bool(args.thermal),
                   "parallelism": (f"env-sharded x{world}; per step ONE {args.gather_algo} of 56 B/env records (written by the step kernel straight into the send buffer) into the "
                                   f"global record array, pipelined under the next kernel; the timed loop hands out the learner's typed views "
                                   f"(obs [N,12], reward, terminated, truncated) -- no copies") if world > 1 else "single GPU"},
        "host": host,
        "timing": {"value_is": f"C-ABI step (HipBackend.step -> stg_step_many): host wall time of a block of exactly `steps` steps between "
                               f"synchronizes (+ barriers, max over ranks); {DEFAULT_BLOCKS} such blocks are timed and the MEDIAN one is "
                               f"reported (block_walls_ms lists all, block_used which)",
                   "api_ms_per_step_is": "the same steps through the public SpinTorqueVecEnv.step() with [N,2] actions",
                   "retime_rule": "a block is DISCARDED and replaced (at most twice) only if the container's cgroup cpu.stat shows "
                                  "nr_throttled increasing during it (CFS bandwidth throttling of the whole container); the measured "
                                  "times never decide what is kept",
                   "gc": "Python collector disabled inside a timed block"},
        "roofline": roofline_step(meas, n_local, args.solver, False, row("headline"), pmc_src),
    }
    if world > 1:
        out["ranks"] = ranks
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
        # CPU baseline beside the N > 1 number too: rank 0 times the oracle on the host cores AFTER the timed region while the other
        # ranks sleep on a key of the rendezvous store (no collective is pending meanwhile, nobody spins on a GPU queue)
        out["cpu_baseline"] = None
        if args.cpu_baseline:
            store = None
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:       # noqa: BLE001 -- private accessor: fall back to the barrier below
                pass
            if rank == 0:
                out["cpu_baseline"] = cpu_baseline(args.solver, args.thermal, args.cpu_seconds)
                if store is not None:
                    store.set("stg_bench_cpu_baseline_done", "1")
            elif store is not None:
                from datetime import timedelta
                try:
                    store.wait(["stg_bench_cpu_baseline_done"], timedelta(seconds=4 * args.cpu_seconds + 300))
                except Exception:   # noqa: BLE001
                    pass
            dist.barrier()
        if rank == 0:
            # the in-place all-gather and the point-to-point exchange below have never run over RCCL with N > 1 (one-GPU build
            # boxes): a safety copy of the measured line goes to stderr first, in case one of them does not come back
            print("bench.py (before the exchange sweep): " + json.dumps(dict(out, **block_report_of(meas))), file=sys.stderr, flush=True)
        # ... and a watchdog on every rank: if the sweep hangs, rank 0 still prints the measured line (with the sweep marked as not
        # finished, `error` set) and every rank leaves with exit code 3 -- a wedged exchange is a failure, not a clean run
        def _sweep_timed_out():
            if rank == 0:
                msg = f"the exchange sweep did not finish within {args.sweep_timeout:g} s and was abandoned (exit code 3)"
                out["gather_only"]["all_algos_ms"] = {"error": msg}
                out["error"] = msg
                out.update(block_report_of(meas, order=("block_throttled", "block_used", "block_walls_ms", "blocks_timed")))
                print(json.dumps(out), flush=True)
            sys.stderr.flush()
            os._exit(3)
        import threading
        dog = threading.Timer(args.sweep_timeout, _sweep_timed_out)
        dog.daemon = True
        dog.start()
        sweep = gather_only_all_algos(n_local, args.solver, args.thermal, args.steps, rank, world, local_rank)
        dog.cancel()
        out["gather_only"]["all_algos_ms"] = sweep
    if rank == 0 and world == 1 and args.also:
        also = []
        for key, spec in specs[1:]:
            m = run_row(spec, 0, 1, local_rank)
            st = spec["steps"]
            if spec["kind"] == "step":
                also.append({"workload": key, "value": round(spec["n"] * st / m["wall_s"], 1), "unit": "env-steps/s",
                             "ms_per_step": round(m["wall_s"] / st * 1e3, 4),
                             "api_ms_per_step": round(m["api_ms_per_step"], 4) if m.get("api_ms_per_step") else None,
                             "roofline": roofline_step(m, spec["n"], spec["solver"], spec["mixed"], row(key), pmc_src, per_env=spec["per_env"]),
                             **block_report_of(m)})
            elif spec["kind"] == "short":
                also.append({"workload": m["workload"], "value": round(m["n"] * m["K"] * st / m["wall_s"], 1), "unit": "env-steps/s",
                             "ms_per_step": round(m["wall_s"] / st / m["K"] * 1e3, 5),
                             "roofline": roofline_hbm(m, m["bytes_per_launch"], row(key), pmc_src,
                                                      {"work_units_per_env_step": round(m["work_units"] / max(m["env_steps"], 1), 2),
                                                       "noop_frac": round(m["noop_steps"] / max(m["env_steps"], 1), 6)}),
                             **block_report_of(m)})
            else:
                also.append({"workload": m["workload"], "value": round(m["n"] * st / m["wall_s"], 1), "unit": "array-steps/s",
                             "ms_per_step": round(m["wall_s"] / st * 1e3, 4),
                             "roofline": roofline_hbm(m, m["bytes_per_unit"] * m["n"], row(key), pmc_src,
                                                      {"bytes_per_array_step": m["bytes_per_unit"]}),
                             **block_report_of(m)})
        out["also"] = also
    if rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.solver, args.thermal, args.cpu_seconds) if args.cpu_baseline else None
    # the headline's timing record goes LAST (the driver's log keeps the tail of the line)
    out.update(block_report_of(meas, order=("block_throttled", "block_used", "block_walls_ms", "blocks_timed")))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def block_report_of(m, order=("blocks_timed", "block_walls_ms", "block_throttled", "block_used")):
    return {k: m[k] for k in order}


if __name__ == "__main__":
    main()
