"""Harness parity: the reference's own ways of measuring this path, on the GPU implementation.

  benchmark_physics_simulation ... `spin-torque-gym benchmark` (cli.py:308-342): random actions from the action
                                   space, reset on done, steps per second of ONE environment
  benchmark_vector_env ........... the same loop over a SpinTorqueVecEnv (what the reference cannot do)
  health_report .................. the shape of EnvironmentMonitor.get_health_report (utils/monitoring.py:180-229),
                                   fed by the library's on-device counters instead of host-side bookkeeping
"""
import time
from typing import Any, Dict

import torch

from .envs import SpinTorqueEnv, SpinTorqueVecEnv


def benchmark_physics_simulation(steps: int = 100, **env_kwargs) -> Dict[str, float]:
    env = SpinTorqueEnv(**env_kwargs)
    env.action_space.seed(0) if hasattr(env.action_space, "seed") else None
    env.reset(seed=0)
    for _ in range(10):                                   # warm up (cli.py:313-318)
        _, _, done, truncated, _ = env.step(env.action_space.sample())
        if done or truncated:
            env.reset()
    t0 = time.time()
    env.reset()
    for _ in range(steps):
        _, _, done, truncated, _ = env.step(env.action_space.sample())
        if done or truncated:
            env.reset()
    dt = time.time() - t0
    env.close()
    return {"physics_steps": steps, "total_time": dt, "steps_per_second": steps / dt}


def benchmark_vector_env(num_envs: int = 65536, steps: int = 20, **env_kwargs) -> Dict[str, float]:
    env = SpinTorqueVecEnv(num_envs, autoreset=True, **env_kwargs)
    env.reset(seed=0)
    dev = env.backend.device
    lo = torch.tensor([-env.cfg.max_current, 0.0], device=dev)
    hi = torch.tensor([env.cfg.max_current, env.cfg.max_duration], device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    acts = lo + (hi - lo) * torch.rand((steps + 2, num_envs, 2), generator=g, device=dev)
    for k in range(2):
        env.step(acts[k])
    torch.cuda.synchronize(dev)
    t0 = time.time()
    for k in range(steps):
        env.step(acts[k + 2])
    torch.cuda.synchronize(dev)
    dt = time.time() - t0
    rep = health_report(env)
    env.close()
    return {"num_envs": num_envs, "vector_steps": steps, "total_time": dt, "env_steps_per_second": num_envs * steps / dt,
            "health": rep}


def health_report(env) -> Dict[str, Any]:
    """EnvironmentMonitor.get_health_report-shaped dict from the on-device counters of a SpinTorqueVecEnv / SpinTorqueEnv."""
    vec = env._vec if isinstance(env, SpinTorqueEnv) else env
    c = vec.backend.counters()
    steps = max(c["env_steps"], 1)
    noop_rate = c["noop_steps"] / steps
    issues = []
    if noop_rate > 0.05:        # the reference's max_solver_timeout_rate threshold (monitoring.py:60-66)
        issues.append(f"High solver failure rate: {noop_rate:.2%}")
    status = "HEALTHY" if not issues else "WARNING"
    return {"timestamp": time.time(), "health_status": status, "health_issues": issues,
            "performance_metrics": {"total_steps": c["env_steps"], "solver_work_units": c["work_units"],
                                    "avg_work_units_per_step": c["work_units"] / steps,
                                    "solver_failure_rate": noop_rate, "error_rate": 0.0},
            "recent_performance": {}}
