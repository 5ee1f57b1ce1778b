"""Parity at the BENCHMARKED sizes, for the BENCHMARKED kernel instantiations (`pytest -m gpu`).

bench.py checks nothing, so every configuration it times is run here at its own launch size and compared with the
oracle on contiguous 64-env slices spread over the batch (first / last tile, tile and XCD-group boundaries, a slice
straddling two tiles); the oracle backend of a slice is keyed with the slice's global env index (`env_id0`), exactly as a
multi-GPU shard is.  On top of that, the scheduling options (duration-sorted lane schedule, producer/consumer wavefront
pairs) and the partition of the batch must not change a single bit at that size.

  cfg 3 headline .. RK45, thermal on, 65 536 envs, autoreset: stg_step_kernel<2,true,false,true,false,float,true,1>, the
                    `nwg == 1024` slot map of the wave-specialised launch (csrc/stg_kernels.hpp: stg_slot_block)
  cfg 4 ........... 262 144 mixed STT/SOT/VCMA envs, RK4, reference RHS and torque_model='device': MULTI x WGW = 4
  cfg 2 ........... RK45, T = 0 K, 4 096 envs: every env against the oracle
  cfg 2a .......... 1 048 576 envs, one 1 ps DP5 step per pulse, step_many K = 8 against 8 x step and the oracle
Reference semantics: envs/spin_torque_env.py:310-407 (step), :250-308 (reset).  Tolerances as in test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

from conftest import stt_default_params

pytestmark = pytest.mark.gpu

TOL_RK4 = 1e-10
TOL_RK45 = 1e-8
SLICE = 64


@pytest.fixture(scope="module")
def stg():
    import spin_torque_gym_amd as s
    assert torch.cuda.is_available(), "these tests need the GPU"
    return s


def _inputs(n, seed, jmax=2e6, tlo=1e-10, thi=1e-9, steps=2):
    """bench.py's input distribution (make_actions): J ~ U[-jmax, jmax], pulse ~ U[tlo, thi] as float32; m0 uniform on the
    sphere, targets +-z."""
    rng = np.random.default_rng(seed)
    v = rng.normal(0, 1, (n, 3))
    m0 = v / np.linalg.norm(v, axis=1, keepdims=True)
    tgt = np.where(rng.integers(0, 2, (n, 1)) == 0, 1.0, -1.0) * np.array([[0.0, 0.0, 1.0]])
    acts = np.empty((steps, n, 2), dtype=np.float32)
    acts[..., 0] = rng.uniform(-jmax, jmax, (steps, n))
    acts[..., 1] = rng.uniform(tlo, thi, (steps, n))
    return m0, tgt, acts


def _slice_starts(n):
    """64-env slices spread over the tiles (4096 envs) and XCD groups of the schedule: both ends, the first tile boundary
    (a slice straddling tiles 0 and 1), odd tiles in the middle, the last complete group of 8 tiles."""
    tile = 4096
    cand = [0, tile - 32, tile + 7 * 64, 3 * tile + 1024, n // 2 - 64, n // 2 + tile + 192, (n // (8 * tile)) * 8 * tile - 2 * 64,
            n - 5 * tile + 320, n - tile - 64, n - 64]
    out = []
    for s in cand:
        s = int(min(max(s, 0), n - SLICE))
        if all(abs(s - o) >= SLICE for o in out):
            out.append(s)
    return out


def _run_hip(stg, n, m0, tgt, acts, cls=None, keep=None, **kw):
    """Steps the HIP env through `acts`; returns the device tensors of every step (cloned) -- `keep` (index tensor on the
    device) restricts what is kept to those envs (the 1M-env case)."""
    env = stg.SpinTorqueVecEnv(n, diagnostics=True, class_index=cls, **kw)
    env.reset(options={"initial_state": m0, "target_state": tgt})
    sel = (lambda t: t.index_select(-1, keep)) if keep is not None else (lambda t: t)
    rec = []
    for a in acts:
        o, r, te, tr, info = env.step(torch.from_numpy(a))
        st = env.get_state()
        d = dict(obs=sel(o.t()).clone(), reward=sel(info["reward_f64"]).clone(), term=sel(te).clone(), trunc=sel(tr).clone(),
                 status=sel(info["status"]).clone(), energy=sel(info["energy"]).clone(), m=sel(st["m"]).clone(),
                 step_count=sel(st["step_count"]).clone(), reward32=sel(r).clone(), etot=sel(st["total_energy"]).clone())
        if kw.get("autoreset"):
            d["final_obs"] = sel(info["final_obs"].t()).clone()
        rec.append(d)
    counters = env.backend.counters()
    env.close()
    return rec, counters


DEFAULT_FORM_KEYS = ("obs", "reward32", "term", "trunc", "status", "m", "step_count", "etot", "final_obs")


def _assert_benchmarked_form_same_bits(stg, hip_rec, counters, n, m0, tgt, acts, tag, cls=None, keep=None, **kw):
    """VERDICT r3 item 3 / ADVICE r3: bench.py times the env's DEFAULT form -- diagnostics off: the C-ABI's optional fp64-reward /
    energy / status-array pointers are NULL, the step writes the 56-byte records (+ terminal observations) only -- while the oracle
    comparisons above run with diagnostics=True.  Same launch, same kernel instantiation: obs / fp32 reward / terminated /
    truncated / the records' status byte / final_obs and the state (m, step count, accumulated energy) must be bit-identical, and the
    on-device work counters equal.  `hip_rec`: what _run_hip returned for the same inputs."""
    env = stg.SpinTorqueVecEnv(n, class_index=cls, **kw)
    assert env.diagnostics is False and env.backend.reward64 is None and env.backend.energy is None
    env.reset(options={"initial_state": m0, "target_state": tgt})
    sel = (lambda t: t.index_select(-1, keep)) if keep is not None else (lambda t: t)
    for k, a in enumerate(acts):
        o, r, te, tr, info = env.step(torch.from_numpy(a))
        assert "reward_f64" not in info and "energy" not in info
        st = env.get_state()
        d = dict(obs=sel(o.t()), reward32=sel(r), term=sel(te), trunc=sel(tr), status=sel(info["status"]), m=sel(st["m"]),
                 step_count=sel(st["step_count"]), etot=sel(st["total_energy"]))
        if kw.get("autoreset"):
            d["final_obs"] = sel(info["final_obs"].t())
        for key, v in d.items():
            # (final_obs rows of envs that did not end hold whatever an earlier step left there: compared where the episode ended)
            if key == "final_obs":
                ended = (d["term"] | d["trunc"])
                assert torch.equal(v[:, ended], hip_rec[k][key][:, ended]), (tag, "diagnostics off", k, key)
            else:
                assert torch.equal(v, hip_rec[k][key]), (tag, "diagnostics off", k, key, int((v != hip_rec[k][key]).sum()))
    assert env.backend.counters() == counters, (tag, "diagnostics off: work counters")
    env.close()


def _run_oracle_slice(stg, s0, m0, tgt, acts, cls=None, **kw):
    from helpers import OracleBackend
    sl = slice(s0, s0 + SLICE)
    env = stg.SpinTorqueVecEnv(SLICE, diagnostics=True, class_index=None if cls is None else cls[sl], env_id0=s0, backend=OracleBackend, **kw)
    env.reset(options={"initial_state": m0[sl], "target_state": tgt[sl]})
    rec = []
    for a in acts:
        o, r, te, tr, info = env.step(torch.from_numpy(a[sl]))
        st = env.get_state()
        d = dict(obs=o.t().clone(), reward=info["reward_f64"].clone(), term=te.clone(), trunc=tr.clone(),
                 status=info["status"].clone(), energy=info["energy"].clone(), m=st["m"].clone(),
                 step_count=st["step_count"].clone())
        if kw.get("autoreset"):
            d["final_obs"] = info["final_obs"].t().clone()
        rec.append(d)
    env.close()
    return rec


def _cmp_slice(hip_rec, ora_rec, cols, tol_m, tag):
    """hip_rec: per-step dicts of device tensors over the whole batch (or the kept envs); cols: where this slice's envs
    sit in them.  Envs that were auto-reset carry a state redrawn from fp32 device normals (1e-7 from libm's): their
    later steps are compared loosely, everything before and including the reset step exactly as tightly as the rest."""
    worst = 0.0
    redrawn = np.zeros(SLICE, dtype=bool)
    for k, (h, o) in enumerate(zip(hip_rec, ora_rec)):
        g = lambda key: h[key][..., cols].cpu().numpy()
        clean = ~redrawn
        assert np.array_equal(g("status")[clean], o["status"].numpy()[clean]), (tag, k)
        assert np.array_equal(g("term")[clean], o["term"].numpy()[clean]) and np.array_equal(g("trunc")[clean], o["trunc"].numpy()[clean]), (tag, k)
        done = (o["term"].numpy() | o["trunc"].numpy()).astype(bool)
        ended = done & clean if "final_obs" in h else np.zeros(SLICE, dtype=bool)
        keep = clean & ~ended                       # envs whose state after this step is the integrated one
        dm = np.abs(g("m") - o["m"].numpy())
        if keep.any():
            worst = max(worst, dm[:, keep].max())
            assert dm[:, keep].max() <= tol_m, (tag, k, dm[:, keep].max())
            assert np.allclose(g("obs")[:, keep], o["obs"].numpy()[:, keep], rtol=3e-7, atol=max(1e-12, 10 * tol_m)), (tag, k)
        assert np.allclose(g("reward")[clean], o["reward"].numpy()[clean], rtol=1e-10, atol=max(1e-12, 10 * tol_m)), (tag, k)
        assert np.allclose(g("energy")[clean], o["energy"].numpy()[clean], rtol=max(1e-12, 10 * tol_m), atol=0), (tag, k)
        if ended.any():
            # the terminal observation is the integrated state's; the redrawn state agrees to the fp32 normals' rounding
            assert np.allclose(g("final_obs")[:, ended], o["final_obs"].numpy()[:, ended], rtol=3e-7, atol=max(1e-12, 10 * tol_m)), (tag, k)
            assert dm[:, ended].max() < 2e-6, (tag, k, dm[:, ended].max())
            assert np.array_equal(g("step_count")[ended], np.zeros(int(ended.sum()), dtype=np.int32))
        if redrawn.any():
            assert dm[:, redrawn & ~done].max(initial=0.0) < 1e-3, (tag, k)
        redrawn |= ended
    return worst


def _assert_same_bits(a, b, tag):
    for k, (x, y) in enumerate(zip(a, b)):
        for key in x:
            assert torch.equal(x[key], y[key]), (tag, k, key, int((x[key] != y[key]).sum()))


# ------------------------------------------------------------------------------------------------------------------
# cfg 3: the headline kernel at its own size
# ------------------------------------------------------------------------------------------------------------------
def test_cfg3_headline_rk45_thermal_65536_vs_oracle_slices(stg):
    n = 65536
    m0, tgt, acts = _inputs(n, seed=1234, steps=2)
    kw = dict(device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=True, temperature=300.0,
              solver="rk45", seed=1234, autoreset=True)
    hip, c = _run_hip(stg, n, m0, tgt, acts, **kw)          # automatic schedule: sorted, wave-specialised, nwg == 1024
    assert c["env_steps"] == 2 * n and c["noop_steps"] == 0
    assert 400 < c["work_units"] / c["env_steps"] < 900      # ~<T>/max_step x 1.2 attempts: no solve was skipped
    worst = 0.0
    for s0 in _slice_starts(n):
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, **kw)
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), TOL_RK45, ("cfg3", s0)))
    print("cfg3 headline (rk45, thermal, 65536): worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "cfg3 headline", **kw)      # what bench.py times
    # scheduling options change nothing at this size: producer/consumer pairs off, identity lane schedule, both
    for opt in (dict(wave_spec=False), dict(lane_sort=False), dict(wave_spec=False, lane_sort=False)):
        other, c2 = _run_hip(stg, n, m0, tgt, acts, **kw, **opt)
        _assert_same_bits(hip, other, ("cfg3", opt))
        assert c2 == c
    # determinism and partition invariance: a repeated run, and two half-size contexts with env_id0
    again, _ = _run_hip(stg, n, m0, tgt, acts, **kw)
    _assert_same_bits(hip, again, "cfg3 repeat")
    h = n // 2
    lo, _ = _run_hip(stg, h, m0[:h], tgt[:h], acts[:, :h], **kw)
    hi, _ = _run_hip(stg, h, m0[h:], tgt[h:], acts[:, h:], env_id0=h, **kw)
    for k in range(len(hip)):
        for key in hip[k]:
            assert torch.equal(hip[k][key], torch.cat([lo[k][key], hi[k][key]], dim=-1)), ("cfg3 halves", k, key)
    norm = torch.linalg.norm(hip[-1]["m"], dim=0)
    assert torch.all(torch.abs(norm - 1) < 1e-14)


def test_cfg5_shard_rk45_thermal_131072_vs_oracle_slices(stg):
    """One cfg5 shard (1 048 576 envs over 8 GPUs = 131 072 per GPU) of the headline workload, keyed as rank 3's shard
    (env_id0 = 3 x 131 072).  Since round 4 this launch takes the lane-refill kernel (1024 persistent wavefronts sharing one global
    queue; with the thermal field from 98 305 envs) -- compared below, bit for bit, with the one-env-per-lane launch it replaced:
    the 4-wavefront thermal RK45 kernel without wave specialisation, two rounds of workgroups
    per CU -- the boustrophedon order of the sorted schedule (csrc/stg_kernels.hpp: stg_slot_block)."""
    n, id0 = 131072, 3 * 131072
    m0, tgt, acts = _inputs(n, seed=77, steps=2)
    kw = dict(device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=True, temperature=300.0,
              solver="rk45", seed=1234, autoreset=True)
    hip, c = _run_hip(stg, n, m0, tgt, acts, env_id0=id0, **kw)
    assert c["env_steps"] == 2 * n and c["noop_steps"] == 0
    worst = 0.0
    for s0 in _slice_starts(n)[::2]:
        from helpers import OracleBackend
        sl = slice(s0, s0 + SLICE)
        env = stg.SpinTorqueVecEnv(SLICE, diagnostics=True, env_id0=id0 + s0, backend=OracleBackend, **kw)
        env.reset(options={"initial_state": m0[sl], "target_state": tgt[sl]})
        ora = []
        for a in acts:
            o, r, te, tr, info = env.step(torch.from_numpy(a[sl]))
            st = env.get_state()
            ora.append(dict(obs=o.t().clone(), reward=info["reward_f64"].clone(), term=te.clone(), trunc=tr.clone(),
                            status=info["status"].clone(), energy=info["energy"].clone(), m=st["m"].clone(),
                            step_count=st["step_count"].clone(), final_obs=info["final_obs"].t().clone()))
        env.close()
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), TOL_RK45, ("cfg5 shard", s0)))
    print("cfg5 shard (rk45, thermal, 131072, env_id0 = 393216): worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "cfg5 shard", env_id0=id0, **kw)
    for variant in (dict(lane_sort=False), dict(lane_refill=False), dict(lane_refill=False, lane_sort=False)):
        other, c2 = _run_hip(stg, n, m0, tgt, acts, env_id0=id0, **kw, **variant)
        _assert_same_bits(hip, other, ("cfg5 shard", variant))
        assert c2 == c, (variant, c2, c)


def test_cfg3_rk4_thermal_65536_vs_oracle_slices(stg):
    """The env's own solver at the headline size (bench.py `also`: cfg3 rk4): RK4 producer/consumer kernel with the
    handshake-word protocol, nwg == 1024 slot map."""
    n = 65536
    m0, tgt, acts = _inputs(n, seed=99, steps=2)
    kw = dict(device_params=stt_default_params(volume=8.75e-11), include_thermal_fluctuations=True, temperature=300.0,
              solver="rk4", seed=1234, autoreset=True)
    hip, c = _run_hip(stg, n, m0, tgt, acts, **kw)
    assert c["env_steps"] == 2 * n
    worst = 0.0
    for s0 in _slice_starts(n)[::2]:
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, **kw)
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), 1e-9, ("cfg3-rk4", s0)))
    print("cfg3 rk4 thermal 65536: worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "cfg3-rk4", **kw)
    other, _ = _run_hip(stg, n, m0, tgt, acts, **kw, wave_spec=False, lane_sort=False)
    _assert_same_bits(hip, other, "cfg3-rk4 options")


# ------------------------------------------------------------------------------------------------------------------
# cfg 4: 262 144 mixed STT/SOT/VCMA envs (MULTI x 4-wavefront workgroups)
# ------------------------------------------------------------------------------------------------------------------
def _cfg4_kwargs(stg, torque_model):
    fac = stg.DeviceFactory()
    vol = 8.75e-11
    stt = fac.get_default_parameters("stt_mram"); stt["volume"] = vol
    sot = fac.get_default_parameters("sot_mram"); sot.update(polarization=0.7, volume=vol)
    vc = fac.get_default_parameters("vcma_mram"); vc.update(polarization=0.7, volume=vol)
    return dict(device_type=["stt_mram", "sot_mram", "vcma_mram"], device_params=[stt, sot, vc],
                include_thermal_fluctuations=False, temperature=300.0, solver="rk4", seed=1234, autoreset=True,
                torque_model=torque_model)


@pytest.mark.parametrize("torque_model", ["reference", "device"])
def test_cfg4_mixed_262144_vs_oracle_slices(stg, torque_model):
    """bench.py's cfg4 rows exactly (run_config(mixed=True)): class = env index mod 3, factory defaults of each type with
    polarization 0.7 and the rescaled volume."""
    n = 262144
    cls = (np.arange(n) % 3).astype(np.uint8)
    m0, tgt, acts = _inputs(n, seed=4321, steps=2)
    kw = _cfg4_kwargs(stg, torque_model)
    hip, c = _run_hip(stg, n, m0, tgt, acts, cls=cls, **kw)
    assert c["env_steps"] == 2 * n
    worst = 0.0
    for s0 in _slice_starts(n):
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, cls=cls, **kw)
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), TOL_RK4, ("cfg4", torque_model, s0)))
    print(f"cfg4 ({torque_model}, 262144 mixed): worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, ("cfg4", torque_model), cls=cls, **kw)
    other, c2 = _run_hip(stg, n, m0, tgt, acts, cls=cls, **kw, lane_sort=False)
    _assert_same_bits(hip, other, ("cfg4 lane_sort off", torque_model))
    assert c2 == c
    # a random class assignment (type-mixed wavefronts whatever the schedule does) at the same size
    cls_r = np.random.default_rng(7).integers(0, 3, n).astype(np.uint8)
    hip_r, _ = _run_hip(stg, n, m0, tgt, acts[:1], cls=cls_r, **kw)
    for s0 in _slice_starts(n)[1::3]:
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts[:1], cls=cls_r, **kw)
        _cmp_slice(hip_r, ora, slice(s0, s0 + SLICE), TOL_RK4, ("cfg4 random classes", torque_model, s0))


def test_cfg4_per_env_parameters_262144_vs_oracle_slices(stg):
    """bench.py's per-env-parameter cfg4 row exactly (run_config(mixed=True, per_env=True)): 262 144 mixed STT/SOT/VCMA envs,
    every env with its own parameter record (stg_set_params_per_env; SURVEY 8d prices it at 272 B/env-step), built by
    bench.mixed_kwargs / bench.per_env_variation.  The oracle of a 64-env slice gets the same envs as 64 device classes (one
    dict per env: its base type's dict with the env's overrides).  Lane sort on/off: identical bits."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = 262144
    m0, tgt, acts = _inputs(n, seed=987, steps=2)
    mk, cls_t = bench.mixed_kwargs("rk4", n, per_env=True)
    cls = cls_t.numpy()
    kw = dict(mk, include_thermal_fluctuations=False, temperature=300.0, solver="rk4", seed=1234, autoreset=True)
    hip, c = _run_hip(stg, n, m0, tgt, acts, cls=cls, **kw)
    assert c["env_steps"] == 2 * n and c["noop_steps"] == 0
    ov = kw["per_env_params"]
    okw = {k: v for k, v in kw.items() if k not in ("per_env_params", "device_type", "device_params")}
    worst = 0.0
    for s0 in _slice_starts(n):
        types, dicts = [], []
        for i in range(s0, s0 + SLICE):
            d = dict(kw["device_params"][cls[i]])
            for key, arr in ov.items():
                d[key] = float(arr[i])
            types.append(kw["device_type"][cls[i]])
            dicts.append(d)
        # (the oracle env of the slice: class j = env s0 + j)
        full_cls = np.zeros(n, dtype=np.uint8)
        full_cls[s0:s0 + SLICE] = np.arange(SLICE, dtype=np.uint8)
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, cls=full_cls, device_type=types, device_params=dicts, **okw)
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), TOL_RK4, ("cfg4 per-env", s0)))
    print("cfg4 per-env (262144 mixed, own record per env): worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "cfg4 per-env", cls=cls, **kw)
    other, c2 = _run_hip(stg, n, m0, tgt, acts, cls=cls, **kw, lane_sort=False)
    _assert_same_bits(hip, other, "cfg4 per-env lane_sort off")
    assert c2 == c


# ------------------------------------------------------------------------------------------------------------------
# lane refill of the RK45 step (csrc/stg_kernels.hpp: stg_step_refill_kernel: persistent wavefronts on one global queue), automatic above
# 131 072 envs (with the thermal field: above 98 304)
# ------------------------------------------------------------------------------------------------------------------
def test_lane_refill_rk45_thermal_262144_vs_oracle_slices_and_one_env_per_lane(stg):
    """VERDICT r2 item 2.  The headline workload at 262 144 envs takes the lane-refill kernel by default (4 envs per lane: a
    lane that has finished its env takes the next one of its wavefront's queue).  Per-env arithmetic is that of the
    one-env-per-lane kernel, so: (i) oracle slices as everywhere else, (ii) identical bits with lane_refill=False, with 2 and
    8 envs per lane, and with the identity lane schedule, (iii) identical on-device work counters."""
    n = 262144
    m0, tgt, acts = _inputs(n, seed=31, steps=2)
    kw = dict(device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=True, temperature=300.0, solver="rk45",
              seed=1234, autoreset=True)
    hip, c = _run_hip(stg, n, m0, tgt, acts, **kw)                       # automatic: refill, 4 envs per lane
    assert c["env_steps"] == 2 * n and c["noop_steps"] == 0
    worst = 0.0
    for s0 in _slice_starts(n):
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, **kw)
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), TOL_RK45, ("refill 262144", s0)))
    print("lane refill (262144, rk45 + thermal): worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "refill 262144", **kw)
    for variant in (dict(lane_refill=False), dict(lane_refill=2), dict(lane_refill=8), dict(lane_sort=False)):
        other, c2 = _run_hip(stg, n, m0, tgt, acts, **kw, **variant)
        _assert_same_bits(hip, other, ("refill variant", variant))
        assert c2 == c, (variant, c2, c)


def test_lane_refill_rk45_thermal_1048576_vs_oracle_slices(stg):
    """bench.py's "cfg5 on ONE GPU" row: 1 048 576 envs, RK45 + thermal, automatic lane refill with 2048 queues of 8 envs per lane:
    oracle slices, and the same bits as the one-env-per-lane launch."""
    n = 1048576
    m0, tgt, acts = _inputs(n, seed=77, steps=1)
    kw = dict(device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=True, temperature=300.0, solver="rk45",
              seed=1234, autoreset=True)
    starts = _slice_starts(n)
    keep = torch.cat([torch.arange(s0, s0 + SLICE) for s0 in starts]).cuda()
    hip, c = _run_hip(stg, n, m0, tgt, acts, keep=keep, **kw)
    assert c["env_steps"] == n and c["noop_steps"] == 0
    worst = 0.0
    for j, s0 in enumerate(starts):
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, **kw)
        worst = max(worst, _cmp_slice(hip, ora, slice(j * SLICE, (j + 1) * SLICE), TOL_RK45, ("refill 1M", s0)))
    print("lane refill (1048576, rk45 + thermal): worst |dm| vs oracle on slices =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "refill 1M", keep=keep, **kw)
    other, c2 = _run_hip(stg, n, m0, tgt, acts, keep=keep, lane_refill=False, **kw)
    _assert_same_bits(hip, other, "refill 1M vs one env per lane")
    assert c2 == c


@pytest.mark.parametrize("thermal", [False, True])
def test_lane_refill_ragged_sizes_and_class_tables(stg, thermal):
    """Forced refill on sizes that are no multiple of anything (last block, last queue round and last tile incomplete; fewer
    blocks than rounds), with a class table in LDS, T = 0 K and thermal: bit-identical to one env per lane."""
    q = stt_default_params(volume=9.7e-6, damping=0.02, polarization=0.5)
    for n, r in ((200001, 3), (4096 * 5 + 70, 2), (130, 4), (64, 2), (9000, 7)):
        m0, tgt, acts = _inputs(n, seed=n, steps=2, thi=4e-10)
        cls = (np.arange(n) % 2).astype(np.uint8)
        kw = dict(device_type=["stt_mram", "stt_mram"], device_params=[stt_default_params(volume=9.7e-6), q],
                  include_thermal_fluctuations=thermal, solver="rk45", seed=5, autoreset=True, max_steps=2, wave_spec=False)
        base, c0 = _run_hip(stg, n, m0, tgt, acts, cls=cls, lane_refill=False, **kw)
        ref, c1 = _run_hip(stg, n, m0, tgt, acts, cls=cls, lane_refill=r, **kw)
        _assert_same_bits(base, ref, ("refill ragged", n, r, thermal))
        assert c0 == c1


@pytest.mark.parametrize("solver,thermal,sizes", [
    ("rk4", False, (4097, 65600, 66000, 69632, 81920, 100000, 131136, 132000, 165000, 200001, 262145)),
    ("rk4", True, (33000, 40960, 50000, 61440, 65535, 65537, 66000, 73729, 81920, 100000, 132000)),   # (three / four pair workgroups per CU; hybrid)
    ("rk45", True, (32832, 36864, 45000, 60000, 65472, 65537, 65600, 66000, 69633, 77777, 81920, 81921, 100000, 131136, 132000, 165000)),      # wave-specialised / hybrid / one-env-per-lane / refill
    ("rk45", False, (66000, 100000, 132000, 170000)),
])
def test_schedule_covers_every_env_once_at_odd_sizes(stg, solver, thermal, sizes):
    """Round 3 rewrote the lane schedule for batch sizes that are no multiple of 8 tiles (32 768 envs) or of a tile (4096): rank-major
    order over all tiles, the ragged tile as a partly empty one, spread / consecutive ranks by the total workgroup count,
    boustrophedon on the resident rounds, hybrid producer/consumer launch up to 98 304 envs (RK4: 90 112), lane refill above 131 072 (thermal: 98 304).  The
    schedule must stay a permutation: every env stepped exactly once (on-device counter) and every output bit equal to the identity
    schedule's (lane_sort=False, which also switches hybrid and refill queues to the identity order)."""
    vol = 9.7e-6 if solver == "rk45" else 8.75e-11
    for n in sizes:
        m0, tgt, acts = _inputs(n, seed=n, steps=1, thi=3e-10)
        kw = dict(device_params=stt_default_params(volume=vol), include_thermal_fluctuations=thermal, solver=solver, seed=3, autoreset=True,
                  max_steps=1)
        a, ca = _run_hip(stg, n, m0, tgt, acts, **kw)
        b, cb = _run_hip(stg, n, m0, tgt, acts, lane_sort=False, **kw)
        assert ca["env_steps"] == n and ca == cb, (solver, thermal, n, ca, cb)
        _assert_same_bits(a, b, ("schedule", solver, thermal, n))


@pytest.mark.parametrize("solver", ["rk45", "rk4"])
def test_hybrid_launch_with_class_table_skip_done_and_fused_steps(stg, solver):
    """The hybrid wave-specialised launch (65 536 < N <= 81 920 envs, thermal: 1024 two-wavefront workgroups, producer/consumer pairs for
    the longest blocks, two blocks with inline normals in each of the others) with what the other tests of that size range leave out: a
    class table (MULTI kernels), `skip_done` without auto-reset over two steps (finished envs are not integrated), K = 2 fused steps,
    float64 actions.  Every env stepped the same number of times and all bits equal to the identity schedule's (which does not use it)."""
    import torch
    vol = 9.7e-6 if solver == "rk45" else 8.75e-11
    fac = stg.DeviceFactory()
    stt = fac.get_default_parameters("stt_mram"); stt["volume"] = vol
    vc = fac.get_default_parameters("vcma_mram"); vc.update(polarization=0.6, volume=vol * 0.8)
    for n in (65537, 70001, 81920):
        cls = (np.arange(n) % 2).astype(np.uint8)
        m0, tgt, acts = _inputs(n, seed=n + 1, steps=2, thi=2.5e-10)
        kw = dict(device_type=["stt_mram", "vcma_mram"], device_params=[stt, vc], include_thermal_fluctuations=True, solver=solver, seed=5,
                  max_steps=1, skip_done=True, autoreset=False)
        a, ca = _run_hip(stg, n, m0, tgt, acts, cls=cls, **kw)
        b, cb = _run_hip(stg, n, m0, tgt, acts, cls=cls, lane_sort=False, **kw)
        assert ca == cb and ca["env_steps"] == n, (solver, n, ca, cb)          # (second step: every env is done and skipped)
        _assert_same_bits(a, b, ("hybrid class table skip_done", solver, n))
        # K = 2 fused steps, float64 actions
        outs = []
        for ls in (None, False):
            env = stg.SpinTorqueVecEnv(n, diagnostics=True, class_index=cls, lane_sort=ls, **dict(kw, max_steps=100, skip_done=False, autoreset=True))
            env.reset(options={"initial_state": m0, "target_state": tgt})
            o, r, te, tr, info = env.step_many(torch.from_numpy(np.stack(acts).astype(np.float64)))
            outs.append((o.clone(), r.clone(), te.clone(), tr.clone(), env.get_state()["m"].clone()))
            env.close()
        for x, y in zip(*outs):
            assert torch.equal(x, y), (solver, n, "fused")


@pytest.mark.parametrize("thermal", [False, True])
def test_device_physics_schedule_at_ragged_sizes(stg, thermal):
    """The device-physics torque model groups the lanes of a tile by device kind and then renumbers the tile's 64-slot blocks by work
    (plan kernel; since round 4 at every size, kind-pure groups dealt by estimated cost): workgroups stay type-uniform, rank order is longest-first.  A ragged tile's partly
    filled group must keep its place (the step kernel's `slot < N` test): every env stepped exactly once and all bits equal to the identity
    schedule's, at sizes that end inside a block, on a block boundary, on a tile boundary, and with random class assignments."""
    for n, seed in ((4097, 1), (70001, 2), (98304, 3), (98304 + 64 * 7, 4), (100001, 5), (126999, 6), (131072, 7), (150017, 8)):   # (sizes around the round-3 regroup range)
        rng = np.random.default_rng(seed)
        cls = rng.integers(0, 3, n).astype(np.uint8) if seed % 2 else (np.arange(n) % 3).astype(np.uint8)
        m0, tgt, acts = _inputs(n, seed=n, steps=1, thi=4e-10)
        kw = _cfg4_kwargs(stg, "device")
        kw.update(include_thermal_fluctuations=thermal, max_steps=1)
        a, ca = _run_hip(stg, n, m0, tgt, acts, cls=cls, **kw)
        b, cb = _run_hip(stg, n, m0, tgt, acts, cls=cls, lane_sort=False, **kw)
        assert ca["env_steps"] == n and ca == cb, (n, ca, cb)
        _assert_same_bits(a, b, ("device-physics schedule", n, thermal))


# ------------------------------------------------------------------------------------------------------------------
# cfg 2: 4096 envs, T = 0 K, RK45 -- every env
# ------------------------------------------------------------------------------------------------------------------
def test_cfg2_rk45_4096_every_env_vs_oracle(stg):
    from helpers import OracleBackend
    n = 4096
    m0, tgt, acts = _inputs(n, seed=2, steps=2)
    kw = dict(device_params=stt_default_params(volume=9.7e-6), include_thermal_fluctuations=False, solver="rk45", seed=1234,
              autoreset=True)
    hip, c = _run_hip(stg, n, m0, tgt, acts, **kw)
    worst = 0.0
    for s0 in range(0, n, SLICE):
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, **kw)
        worst = max(worst, _cmp_slice(hip, ora, slice(s0, s0 + SLICE), TOL_RK45, ("cfg2", s0)))
    print("cfg2 rk45 4096 (every env): worst |dm| vs oracle =", worst)
    _assert_benchmarked_form_same_bits(stg, hip, c, n, m0, tgt, acts, "cfg2", **kw)
    other, _ = _run_hip(stg, n, m0, tgt, acts, **kw, lane_sort=False)
    _assert_same_bits(hip, other, "cfg2 lane_sort off")


# ------------------------------------------------------------------------------------------------------------------
# cfg 2a: 1 048 576 envs, one 1 ps DP5 step per pulse, K = 8 fused
# ------------------------------------------------------------------------------------------------------------------
def test_cfg2a_1m_envs_step_many_k8_vs_steps_and_oracle(stg):
    """bench.py's run_short_pulse_config: rk45, T = 0 K, default STT parameters, J = 0, every pulse 1 ps, autoreset,
    identity schedule.  (i) step_many(K = 8) == 8 x step, bit for bit, over all 1M envs; (ii) slices against the oracle."""
    n, K = 1048576, 8
    m0, tgt, _ = _inputs(n, seed=8, steps=1)
    acts = np.zeros((K, n, 2), dtype=np.float32)
    acts[..., 1] = 1e-12
    # a few non-trivial pulses so that the fused loop also carries J != 0, longer durations and the action clamp
    rng = np.random.default_rng(5)
    odd = rng.integers(0, n, 4096)
    acts[:, odd, 0] = rng.uniform(-2e-10, 2e-10, (K, len(odd))).astype(np.float32)   # beta J ~ the precession rate: default volume, not stiff
    acts[:, odd, 1] = rng.uniform(1e-12, 4e-12, (K, len(odd))).astype(np.float32)
    kw = dict(solver="rk45", include_thermal_fluctuations=False, seed=1, autoreset=True, lane_sort=False)
    starts = _slice_starts(n)
    keep = torch.cat([torch.arange(s, s + SLICE) for s in starts]).cuda()

    e1 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw)
    e1.reset(options={"initial_state": m0, "target_state": tgt})
    om, rm, tem, trm, im = e1.step_many(torch.from_numpy(acts), out_every=False)
    st1 = e1.get_state()
    c1 = e1.backend.counters()
    e2 = stg.SpinTorqueVecEnv(n, diagnostics=True, **kw)
    e2.reset(options={"initial_state": m0, "target_state": tgt})
    per_step = []
    for k in range(K):
        o, r, te, tr, info = e2.step(torch.from_numpy(acts[k]))
        st = e2.get_state()
        per_step.append(dict(obs=o.t().index_select(-1, keep).clone(), reward=info["reward_f64"].index_select(-1, keep).clone(),
                             term=te.index_select(-1, keep).clone(), trunc=tr.index_select(-1, keep).clone(),
                             status=info["status"].index_select(-1, keep).clone(), energy=info["energy"].index_select(-1, keep).clone(),
                             m=st["m"].index_select(-1, keep).clone(), step_count=st["step_count"].index_select(-1, keep).clone(),
                             final_obs=info["final_obs"].t().index_select(-1, keep).clone()))
    st2 = e2.get_state()
    c2 = e2.backend.counters()
    assert c1 == c2 and c1["env_steps"] == K * n
    for key in ("m", "target", "total_energy", "step_count", "rng_step", "done"):
        assert torch.equal(st1[key], st2[key]), key
    assert torch.equal(om[0], o) and torch.equal(im["reward_f64"][0], info["reward_f64"])
    assert torch.equal(tem[0], te) and torch.equal(trm[0], tr)
    e1.close(); e2.close()
    worst = 0.0
    for j, s0 in enumerate(starts):
        ora = _run_oracle_slice(stg, s0, m0, tgt, acts, **{k: v for k, v in kw.items() if k != "lane_sort"})
        worst = max(worst, _cmp_slice(per_step, ora, slice(j * SLICE, (j + 1) * SLICE), TOL_RK45, ("cfg2a", s0)))
    print("cfg2a (1M envs, K = 8): worst |dm| vs oracle on slices =", worst)


# ------------------------------------------------------------------------------------------------------------------
# cfg 3's statistical gate as SURVEY 8d.3 words it: P(switch) per (J, T_pulse) bin within binomial 3 sigma
# ------------------------------------------------------------------------------------------------------------------
STAT_J = (2e5, 3.5e5, 5e5, 1e6)
STAT_T = (1e-10, 2e-10, 3e-10, 5e-10)
STAT_N = 65536                       # samples per bin, HIP and oracle alike


def _oracle_bins(solver, vol, m0_rows, bins, seed, per_bin=None):
    """The oracle over len(bins) x per_bin envs with its OWN stream key (OpenMP over envs): returns per bin and env the
    status and m_z after one env.step.  m0_rows: [len(bins) * per_bin, 3] initial states (one row per env)."""
    per_bin = STAT_N if per_bin is None else per_bin
    import ctypes as C
    import oracle
    n = len(bins) * per_bin
    p = (oracle.Params * 1)(oracle.make_params(stt_default_params(volume=vol)))
    c = oracle.make_config(solver=solver, thermal=True, seed=seed)
    st = (oracle.EnvState * n)()
    view = np.frombuffer(st, dtype=np.float64).reshape(n, C.sizeof(oracle.EnvState) // 8)
    assert oracle.EnvState.m.offset == 0 and oracle.EnvState.target.offset == 24
    view[:, 0:3] = m0_rows
    view[:, 3:6] = [0.0, 0.0, -1.0]
    a = np.empty((n, 2), dtype=np.float32)
    for b, (J, T) in enumerate(bins):
        a[b * per_bin:(b + 1) * per_bin] = (J, T)
    outs = oracle.env_step_batch(st, a, p, None, c, env_id0=0, n_threads=0)
    raw = np.frombuffer(outs, dtype=np.uint8).reshape(n, C.sizeof(oracle.StepOut))
    return (raw[:, oracle.StepOut.status.offset].reshape(len(bins), per_bin).copy(),
            view[:, 2].reshape(len(bins), per_bin).copy())


def _binomial_gate(p_hip, p_cpu, n_hip, n_cpu, tag):
    """|p_hip - p_cpu| within 3 sigma of the difference of two binomial estimates (pooled p)."""
    pool = (p_hip * n_hip + p_cpu * n_cpu) / (n_hip + n_cpu)
    sigma = np.sqrt(max(pool * (1 - pool), 1e-12) * (1.0 / n_hip + 1.0 / n_cpu))
    z = (p_hip - p_cpu) / sigma
    if pool * (1 - pool) * min(n_hip, n_cpu) < 5:        # (nearly) deterministic bin: at most a handful of envs may differ
        assert abs(p_hip - p_cpu) * min(n_hip, n_cpu) <= 8, (tag, p_hip, p_cpu)
        return 0.0
    assert abs(z) <= 3.0, (tag, p_hip, p_cpu, z)
    return z


def test_cfg3_switching_statistics_grid_independent_streams(stg):
    """SURVEY 8d.3 / BASELINE config 3: with the thermal field on, P(switch) and P(failed solve) per (J, T_pulse) bin must
    match the CPU restatement run with its OWN random stream, within binomial 3 sigma, >= 65 536 samples per bin on both
    sides.  The reference's Brown field carries no 1/sqrt(dt) (simple_solver.py:378-386): the regime in which
    it decides outcomes at all is the torque-dominated small volume V = 1e-28 m^3, where it tips sub-steps between
    "components overflow -> +z -> solve succeeds" and "norm overflows -> zero row -> solve fails" (H3); a 4 x 4 grid there
    spans P(fail) from ~1 % to ~98 %.  One launch of 16 x 65 536 = 1 048 576 envs (bin = block of envs), the env's own
    RK4 solver.  Seeds are fixed, so the test is deterministic; the 3 sigma bound is per bin."""
    bins = [(J, T) for J in STAT_J for T in STAT_T]
    n = len(bins) * STAT_N
    m0 = np.array([0.05, 0.0, 1.0]); m0 /= np.linalg.norm(m0)
    par = stt_default_params(volume=1e-28)
    env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=par, include_thermal_fluctuations=True, solver="rk4", seed=11)
    env.reset(options={"initial_state": m0, "target_state": np.array([0.0, 0.0, -1.0])})
    a = np.empty((n, 2), dtype=np.float32)
    for b, (J, T) in enumerate(bins):
        a[b * STAT_N:(b + 1) * STAT_N] = (J, T)
    _, _, te, tr, info = env.step(torch.from_numpy(a))
    st_h = info["status"].cpu().numpy().reshape(len(bins), STAT_N)
    mz_h = env.get_state()["m"][2].cpu().numpy().reshape(len(bins), STAT_N)
    env.close()
    st_c, mz_c = _oracle_bins("rk4", 1e-28, np.tile(m0, (n, 1)), bins, seed=22)
    zs, stochastic = [], 0
    for b, (J, T) in enumerate(bins):
        f_h, f_c = float((st_h[b] == 1).mean()), float((st_c[b] == 1).mean())
        s_h, s_c = float((mz_h[b] < 0).mean()), float((mz_c[b] < 0).mean())
        print(f"bin J={J:g} T={T:g}: P(fail) HIP {f_h:.4f} CPU {f_c:.4f}   P(switch) HIP {s_h:.4f} CPU {s_c:.4f}")
        zs.append(_binomial_gate(f_h, f_c, STAT_N, STAT_N, ("fail", J, T)))
        zs.append(_binomial_gate(s_h, s_c, STAT_N, STAT_N, ("switch", J, T)))
        stochastic += int(0.01 < f_c < 0.99)
    assert stochastic >= 12                                  # the grid really is in the stochastic regime
    zs = np.array([z for z in zs if z != 0.0])
    assert np.sqrt(np.mean(zs ** 2)) < 1.5, zs               # and the z scores look like unit normals, not like a bias


def test_cfg3_switching_statistics_well_conditioned_volume(stg):
    """The same gate at the well-conditioned volume (8.75e-11 m^3: clean +z -> -z switching, SURVEY headline 3).  The Brown
    field is ~1e-9 of H_k there, so an env's outcome is decided by (m0, J, T); with m0 drawn independently on both sides
    (uniform on the cap m_z > 0.5) P(switch) per bin is a genuine binomial comparison.  RK4 and the RK45 headline solver."""
    bins = [(2e6, 6e-11), (2e6, 1e-10), (1e6, 1e-10), (1e6, 1.5e-10), (5e5, 2.5e-10), (1e6, 6e-10)]
    n = len(bins) * STAT_N

    def cap(seed):
        rng = np.random.default_rng(seed)
        z = rng.uniform(0.5, 1.0, n); ph = rng.uniform(0, 2 * np.pi, n); r = np.sqrt(1 - z * z)
        return np.stack([r * np.cos(ph), r * np.sin(ph), z], axis=1)
    a = np.empty((n, 2), dtype=np.float32)
    for b, (J, T) in enumerate(bins):
        a[b * STAT_N:(b + 1) * STAT_N] = (J, T)
    for solver, vol in (("rk4", 8.75e-11), ("rk45", 9.7e-6)):
        env = stg.SpinTorqueVecEnv(n, diagnostics=True, device_params=stt_default_params(volume=vol), include_thermal_fluctuations=True,
                                   solver=solver, seed=5)
        env.reset(options={"initial_state": cap(1), "target_state": np.array([0.0, 0.0, -1.0])})
        _, _, te, tr, info = env.step(torch.from_numpy(a))
        st_h = info["status"].cpu().numpy().reshape(len(bins), STAT_N)
        mz_h = env.get_state()["m"][2].cpu().numpy().reshape(len(bins), STAT_N)
        env.close()
        n_cpu = 16384 if solver == "rk45" else STAT_N           # (>= 16 384 own-stream oracle samples per bin)
        rows = cap(2)
        rows = np.concatenate([rows[b * STAT_N:b * STAT_N + n_cpu] for b in range(len(bins))])
        st_c, mz_c = _oracle_bins(solver, vol, rows, bins, seed=6, per_bin=n_cpu)
        assert (st_h == 0).all() and (st_c == 0).all()          # P(failed solve) = 0 on both sides
        mixed = 0
        for b, (J, T) in enumerate(bins):
            s_h, s_c = float((mz_h[b] < 0).mean()), float((mz_c[b] < 0).mean())
            print(f"{solver} V={vol:g} J={J:g} T={T:g}: P(switch) HIP {s_h:.4f} CPU {s_c:.4f}")
            _binomial_gate(s_h, s_c, STAT_N, n_cpu, (solver, "switch", J, T))
            mixed += int(0.02 < s_c < 0.98)
        assert mixed >= 4, (solver, mixed)



# ------------------------------------------------------------------------------------------------------------------
# SpinTorqueArray-v0 at the benchmarked size (bench.py run_array_config: 262 144 arrays of 4 x 4 cells)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["individual", "global"])
def test_array_env_262144_vs_oracle_slices(stg, mode):
    """Both array kernels bench.py times (the streaming 'individual' kernel and the LDS-staged general one in 'global' mode),
    262 144 arrays, bench.py's action distribution, two steps: slices of 64 arrays against the oracle, whole batch repeatable."""
    from helpers import OracleArrayBackend
    n, size = 262144, (4, 4)
    ndev = size[0] * size[1]
    rng = np.random.default_rng(31)
    v = rng.normal(0, 1, (n, size[0], size[1], 3))
    init = v / np.linalg.norm(v, axis=-1, keepdims=True)
    a_dim = 2 if mode == "global" else 3
    acts = np.empty((2, n, a_dim), dtype=np.float32)
    if mode == "global":
        acts[..., 0] = rng.uniform(-2e6, 2e6, (2, n)); acts[..., 1] = rng.uniform(-2e6, 2e6, (2, n))
    else:
        acts[..., 0] = rng.uniform(0, ndev, (2, n)); acts[..., 1] = rng.uniform(-2e6, 2e6, (2, n))
        acts[..., 2] = rng.uniform(1e-10, 1e-9, (2, n))
    kw = dict(action_mode=mode, max_steps=10**6, success_threshold=2.0)
    runs = []
    for rep in range(2):
        env = stg.SpinTorqueArrayVecEnv(n, size, seed=3, **kw)
        obs, _ = env.reset(options={"initial_pattern": init})
        rec = [obs.clone()]
        for a in acts:
            obs, r, te, tr, info = env.step(torch.from_numpy(a))
            rec.append((obs.clone(), info["reward_f64"].clone(), te.clone(), tr.clone(), info["energy"].clone(),
                        env.get_state()["pattern"].clone()))
        env.close()
        runs.append(rec)
    assert torch.equal(runs[0][0], runs[1][0])
    for x, y in zip(runs[0][1:], runs[1][1:]):
        assert all(torch.equal(p, q) for p, q in zip(x, y))
    hip = runs[0]
    worst = 0.0
    for s0 in _slice_starts(n)[::2]:
        sl = slice(s0, s0 + SLICE)
        env = stg.SpinTorqueArrayVecEnv(SLICE, size, backend=OracleArrayBackend, **kw)
        obs, _ = env.reset(options={"initial_pattern": init[sl]})
        assert np.allclose(hip[0][sl].cpu().numpy(), obs.numpy(), rtol=2e-7, atol=1e-12)
        for k, a in enumerate(acts):
            obs, r, te, tr, info = env.step(torch.from_numpy(a[sl]))
            h = hip[k + 1]
            pat = env.get_state()["pattern"].numpy()
            d = np.abs(h[5][:, sl].cpu().numpy() - pat).max()
            worst = max(worst, d)
            assert d <= 1e-11, (mode, s0, k, d)
            assert np.allclose(h[0][sl].cpu().numpy(), obs.numpy(), rtol=3e-7, atol=1e-10)
            assert np.allclose(h[1][sl].cpu().numpy(), info["reward_f64"].numpy(), rtol=1e-9, atol=1e-9)
            assert np.array_equal(h[2][sl].cpu().numpy(), te.numpy()) and np.array_equal(h[3][sl].cpu().numpy(), tr.numpy())
            assert np.allclose(h[4][sl].cpu().numpy(), info["energy"].numpy(), rtol=1e-10, atol=0)
        env.close()
    print(f"array env ({mode}, 262144 arrays): worst |dm| vs oracle on slices =", worst)
