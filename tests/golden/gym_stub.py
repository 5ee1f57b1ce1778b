"""Minimal stand-in for the `gymnasium` package (not installed in this image).

Test infrastructure only.  It exists so that ``tests/golden/make_golden.py`` can import
the *unmodified* reference package from ``/root/reference`` and record its outputs as
golden vectors.  It provides exactly the symbols the reference's hot-path modules touch
(SURVEY.md section 7 step 1): ``gymnasium.Env``, ``gymnasium.spaces.{Box,Discrete,Dict}``,
``gymnasium.envs.registration.register`` and ``gymnasium.utils.seeding.np_random``.
"""
import sys
import types

import numpy as np


class Env:
    metadata = {}
    render_mode = None
    _np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, _ = np_random(seed)

    def close(self):
        pass


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low)
        high = np.asarray(high)
        if shape is None:
            shape = np.broadcast(low, high).shape
        super().__init__(tuple(shape), dtype)
        self.low = np.broadcast_to(low, shape).astype(dtype)
        self.high = np.broadcast_to(high, shape).astype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def contains(self, x):
        return 0 <= int(x) < self.n


class Dict(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = dict(spaces)


def np_random(seed=None):
    """Same construction as gymnasium.utils.seeding.np_random: PCG64(SeedSequence(seed))."""
    seed_seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


_registry = {}


def register(id, entry_point=None, max_episode_steps=None, kwargs=None, **_):
    _registry[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=kwargs or {})


def make(id, **kwargs):
    """gymnasium.make without the wrappers: resolves the registered entry point ("module:attr"), merges the registered kwargs
    with the caller's and instantiates.  The spec is left on the env as `.spec` (a dict)."""
    import importlib
    spec = _registry[id]
    ep = spec["entry_point"]
    if isinstance(ep, str):
        mod, attr = ep.split(":")
        ep = getattr(importlib.import_module(mod), attr)
    env = ep(**{**spec["kwargs"], **kwargs})
    env.spec = dict(spec, id=id)
    return env


def install():
    """Register the stub under the name ``gymnasium`` in sys.modules."""
    if "gymnasium" in sys.modules:
        return
    g = types.ModuleType("gymnasium")
    g.Env = Env
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box, spaces.Discrete, spaces.Dict, spaces.Space = Box, Discrete, Dict, Space
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = register
    reg.registry = _registry
    envs.registration = reg
    utils = types.ModuleType("gymnasium.utils")
    seeding = types.ModuleType("gymnasium.utils.seeding")
    seeding.np_random = np_random
    utils.seeding = seeding
    g.spaces, g.envs, g.utils, g.register, g.make = spaces, envs, utils, register, make
    for name, mod in [("gymnasium", g), ("gymnasium.spaces", spaces), ("gymnasium.envs", envs),
                      ("gymnasium.envs.registration", reg), ("gymnasium.utils", utils),
                      ("gymnasium.utils.seeding", seeding)]:
        sys.modules[name] = mod
