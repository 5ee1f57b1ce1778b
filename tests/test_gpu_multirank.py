"""N > 1 on the GPU (`pytest -m gpu`): the HIP backend under ShardedSpinTorqueVecEnv with two live processes.

The build boxes have ONE MI355X, and RCCL refuses two ranks on one GPU, so the two ranks share cuda:0 and exchange over
gloo -- everything else is the product's multi-GPU path (spin_torque_gym_amd/distributed.py): contiguous shards with
`env_id0`, the step kernel writing its 56-byte records straight into the exchange buffer, the global record array, typed
views, and -- with overlap=True -- the side-stream / event protocol that orders the exchange against the step kernels
(gather_begin / gather_end, the double-buffered `_done[k]` waits), here with two HIP contexts stepping real records.
tests/test_dist_gloo.py covers the same host logic on CPU with the oracle as the backend; bench.py --gpus 2 (below) is
the benchmark's own multi-rank form.  Reference semantics: envs/spin_torque_env.py:310-407 (step), :250-308 (reset); the
reference has no distributed path.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import stt_default_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ENVS, STEPS, SLICE = 32768, 3, 64
KW = dict(include_thermal_fluctuations=True, temperature=300.0, solver="rk45", seed=1234, autoreset=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(n, steps):
    rng = np.random.default_rng(2024)
    v = rng.normal(0, 1, (n, 3))
    m0 = v / np.linalg.norm(v, axis=1, keepdims=True)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    acts = np.empty((steps, n, 2), dtype=np.float32)
    acts[..., 0] = rng.uniform(-2e6, 2e6, (steps, n))
    acts[..., 1] = rng.uniform(1e-10, 6e-10, (steps, n))
    return m0, tgt, acts


def _hip_worker(rank, world, port, n, steps, q):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "spin-torque-rl-gym_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from spin_torque_gym_amd.backend import HipBackend, record_views
        from spin_torque_gym_amd.distributed import ShardedSpinTorqueVecEnv, global_views
        m0, tgt, acts = _inputs(n, steps)
        kw = dict(KW, device_params=stt_default_params(volume=9.7e-6))
        opts = {"initial_state": m0, "target_state": tgt}
        runs = {}
        # (i) synchronous form: step(gather=True) returns the global typed views
        env = ShardedSpinTorqueVecEnv(n, **kw)
        assert isinstance(env.local.backend, HipBackend) and env.local.backend.device.type == "cuda"       # the product backend
        assert (env.lo, env.hi) == (rank * n // world, (rank + 1) * n // world) and env.local.env_id0 == env.lo
        obs, _ = env.reset(options=opts)
        rec = [env._glob[env._last].clone()]
        assert torch.equal(obs, record_views(rec[0])[0])
        for k in range(steps):
            obs, r, te, tr, _ = env.step(torch.from_numpy(acts[k]))
            g = env._glob[env._last]
            assert tuple(obs.shape) == (n, 12) and tuple(obs.stride()) == (14, 1) and obs.is_cuda
            assert obs.untyped_storage().data_ptr() == g.untyped_storage().data_ptr()                      # views, no copies
            o2, r2, te2, tr2 = global_views(g)
            assert torch.equal(obs, o2) and torch.equal(r, r2) and torch.equal(te, te2) and torch.equal(tr, tr2)
            rec.append(g.clone())
        st_sync = {k: v.clone() for k, v in env.local.get_state().items()}
        env.close()
        runs["sync"] = rec
        # (ii)-(v) the pipelined form (step k's exchange collected after step k+1 was enqueued) for both exchange algorithms, without
        # and with the side-stream protocol (overlap=True: events order the exchange against the kernels of both record arrays)
        for name, algo, inplace, overlap in (("pipelined", "all_gather", False, None), ("p2p", "p2p", False, None),
                                             ("pipelined overlap", "all_gather", False, True), ("inplace overlap", "all_gather", True, True),
                                             ("p2p overlap", "p2p", False, True)):
            e = ShardedSpinTorqueVecEnv(n, gather_algo=algo, inplace=inplace, overlap=overlap, **kw)
            assert e._overlap is bool(overlap)
            e.reset(options=opts, gather=False)
            got = []
            for k in range(steps):
                e.step(torch.from_numpy(acts[k]), gather=False)
                if k:
                    got.append(e.gather_end(unpack=False).clone())
                e.gather_begin()
            got.append(e.gather_end(unpack=False).clone())
            torch.cuda.synchronize()
            for k in range(steps):
                assert torch.equal(got[k], rec[k + 1]), (name, k, int((got[k] != rec[k + 1]).sum()))
            for key, v in e.local.get_state().items():
                assert torch.equal(v, st_sync[key]), (name, key)
            e.close()
        if rank == 0:
            q.put([x.cpu().numpy().copy() for x in rec])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_hip_ranks_sharded_env_equals_single_process_and_oracle():
    """VERDICT r3 item 2.  Two gloo ranks, both on cuda:0, default HipBackend, ShardedSpinTorqueVecEnv(32768, rk45, thermal,
    autoreset): the gathered global record arrays equal a single-process SpinTorqueVecEnv(32768) BIT FOR BIT for reset + 3 steps,
    in the synchronous form, the pipelined gather_begin / gather_end form, the in-place form and gather_algo='p2p', each also through
    the side-stream / event protocol; two 64-env slices (one per shard) agree with the oracle keyed by the slice's env_id0."""
    import spin_torque_gym_amd as stg
    from helpers import OracleBackend
    from spin_torque_gym_amd.backend import record_views
    assert torch.cuda.is_available()
    n, steps, world = N_ENVS, STEPS, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        rec = [torch.from_numpy(x) for x in q.get(timeout=600)]
    finally:
        for p in procs:
            p.join(timeout=120)
    assert [p.exitcode for p in procs] == [0, 0]
    m0, tgt, acts = _inputs(n, steps)
    kw = dict(KW, device_params=stt_default_params(volume=9.7e-6))
    # one process, one context, all n envs
    env = stg.SpinTorqueVecEnv(n, **kw)
    env.reset(options={"initial_state": m0, "target_state": tgt})
    single = [env.backend.packed.cpu().clone()]
    for k in range(steps):
        env.step(torch.from_numpy(acts[k]))
        single.append(env.backend.packed.cpu().clone())
    assert env.backend.counters()["env_steps"] == steps * n
    env.close()
    for k, (a, b) in enumerate(zip(rec, single)):
        # (bit-identical: the thermal stream and the device-side reset draws are keyed by the GLOBAL env index)
        assert torch.equal(a, b), ("two ranks vs one process", k, int((a != b).sum()))
    # the oracle on one slice per shard, keyed by env_id0
    worst = 0.0
    for s0 in (4096 - 32, n // 2 + 7 * 64):
        sl = slice(s0, s0 + SLICE)
        ora = stg.SpinTorqueVecEnv(SLICE, diagnostics=True, env_id0=s0, backend=OracleBackend, **kw)
        ora.reset(options={"initial_state": m0[sl], "target_state": tgt[sl]})
        redrawn = np.zeros(SLICE, dtype=bool)
        for k in range(steps):
            o, r, te, tr, info = ora.step(torch.from_numpy(acts[k][sl]))
            ho, hr, hte, htr, hst = (t.numpy() for t in record_views(rec[k + 1][sl]))
            clean = ~redrawn
            assert np.array_equal(hte[clean].astype(bool), te.numpy()[clean]) and np.array_equal(htr[clean].astype(bool), tr.numpy()[clean]), (s0, k)
            assert np.array_equal(hst[clean], info["status"].numpy()[clean]), (s0, k)
            ended = (te.numpy() | tr.numpy()) & clean
            keep = clean & ~ended            # (an env that ended holds a state redrawn from fp32 device normals: 1e-7 from libm's)
            d = np.abs(ho[keep] - o.numpy()[keep])
            worst = max(worst, float(d[:, :3].max(initial=0.0)))
            assert np.allclose(ho[keep], o.numpy()[keep], rtol=3e-7, atol=1e-7), (s0, k, d.max())
            assert np.allclose(hr[clean], r.numpy()[clean], rtol=1e-6, atol=1e-7), (s0, k)
            assert np.allclose(ho[ended], o.numpy()[ended], rtol=0, atol=2e-6), (s0, k)
            redrawn |= ended
        ora.close()
    print("two HIP ranks (gloo, cuda:0 shared), 32768 envs rk45 + thermal: worst |obs m - oracle| on slices =", worst)


@pytest.mark.timeout(1200)
def test_bench_py_gpus_2_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with NO launcher around it starts its two ranks itself (before any GPU
    call), prints ONE JSON line with n_gpus 2, both ranks seen, `cpu_baseline` and a non-null `roofline.frac`, and exits 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1",
           "--envs-per-gpu", "16384"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2 and d["ranks"]["backend"] == "gloo"
    assert d["metric"] == "env_steps_per_sec" and d["value"] > 0 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["config"]["global_envs"] == 32768 and d["config"]["envs_per_gpu"] == 16384
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["roofline"]["frac"] is not None and 0 < d["roofline"]["frac"] < 1, d["roofline"]
    assert d["roofline"]["traffic"] is not None and "live" in d["roofline"]["pmc_source"], d["roofline"]["pmc_source"]
    assert d["blocks_timed"] >= 3 and len(d["block_walls_ms"]) == d["blocks_timed"]
    assert set(d["gather_only"]["all_algos_ms"]) == {"all_gather", "all_gather_inplace", "p2p"} and "error" not in d
    out = os.environ.get("STG_BENCH_2RANK_OUT")
    if out:                                                      # (tools/collect_round.sh keeps the line as evidence under profiles/)
        with open(out, "w") as f:
            f.write(lines[0] + "\n")
