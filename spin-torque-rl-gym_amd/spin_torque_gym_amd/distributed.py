"""Multi-GPU: one process per GPU, contiguous shards of independent environments, one collective per step.

The reference has no distributed path at all; this is a new design for the 8-GPU MI355X node.
Environments never interact, so the data path needs no exchange: rank r owns envs [r*n_local, (r+1)*n_local) and
steps them with its own `stg_ctx` (Philox counters use the GLOBAL env index, so results do not depend on the number
of ranks).  The only communication is what an RL learner needs: the step's (obs, reward, terminated, truncated),
54 B/env, packed in one byte buffer and moved by ONE all-gather (RCCL over xGMI when the backend is "nccl"; the
CPU tests run the same code over gloo).  At 131 072 envs per GPU that is 7.1 MB per rank per step.
Actions go the other way: every rank slices its shard out of the global action tensor (a learner that is itself
data-parallel over the same ranks would pass local actions and skip the gather: ``gather=False``).
"""
from typing import Optional

import torch
import torch.distributed as dist

from .backend import PACKED_BYTES_PER_ENV, unpack_step_buffer
from .envs import SpinTorqueVecEnv


def shard_range(n_global: int, world: int, rank: int):
    """Contiguous block partition; the first (n_global % world) ranks take one extra env."""
    base, extra = divmod(n_global, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedSpinTorqueVecEnv:
    """`num_envs` global environments sharded over the ranks of `group` (default: the world group)."""

    def __init__(self, num_envs: int, group: Optional[dist.ProcessGroup] = None, device_index: Optional[int] = None,
                 class_index=None, **env_kwargs):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend 'nccl' = RCCL on ROCm, or 'gloo')")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_envs = int(num_envs)
        self.lo, self.hi = shard_range(self.num_envs, self.world, self.rank)
        self.n_local = self.hi - self.lo
        if self.num_envs % self.world:
            raise ValueError("num_envs must be divisible by the number of ranks (equal shards keep the gather a plain all-gather)")
        if device_index is None:
            device_index = self.rank % max(torch.cuda.device_count(), 1)
        if class_index is not None:
            class_index = torch.as_tensor(class_index)[self.lo:self.hi]
        self.local = SpinTorqueVecEnv(self.n_local, class_index=class_index, device_index=device_index, env_id0=self.lo,
                                      **env_kwargs)
        dev = self.local.backend.packed.device
        self._gathered = torch.zeros(self.world * PACKED_BYTES_PER_ENV * self.n_local, dtype=torch.uint8, device=dev)

    def _gather(self, unpack: bool = True):
        """The single collective of a step.  unpack=False returns the raw [world * 54 * n_local] byte buffer (rank-major;
        `unpack_step_buffer` gives typed views of each rank's slice) without the concatenation copies."""
        packed = self.local.backend.packed
        if packed.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks sharing one GPU): gloo moves host memory
            host = torch.empty(self._gathered.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, packed.cpu(), group=self.group)
            self._gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(self._gathered, packed, group=self.group)
        if not unpack:
            return self._gathered
        n = self.n_local
        parts = [unpack_step_buffer(self._gathered[r * PACKED_BYTES_PER_ENV * n:(r + 1) * PACKED_BYTES_PER_ENV * n], n)
                 for r in range(self.world)]
        obs = torch.cat([p[0] for p in parts], dim=1)       # [12, N_global]
        reward = torch.cat([p[1] for p in parts])
        term = torch.cat([p[2] for p in parts])
        trunc = torch.cat([p[3] for p in parts])
        return obs.t(), reward, term.bool(), trunc.bool()

    def reset(self, seed: Optional[int] = None, options=None, gather: bool = True):
        options = dict(options or {})
        for key in ("initial_state", "target_state", "mask"):
            if options.get(key) is not None:
                v = torch.as_tensor(options[key])
                if v.dim() >= 1 and v.shape[0] == self.num_envs:
                    options[key] = v[self.lo:self.hi]
        obs, info = self.local.reset(seed=seed, options=options)
        if not gather:
            return obs, info
        # reset() fills the packed buffer's obs region; reward/done regions keep their previous content
        return self._gather()[0], info

    def step(self, actions, gather: bool = True, actions_are_local: bool = False):
        a = torch.as_tensor(actions)
        if not actions_are_local:
            a = a[self.lo:self.hi]
        out = self.local.step(a)
        if not gather:
            return out
        obs, reward, term, trunc = self._gather()
        return obs, reward, term, trunc, {}

    def close(self):
        self.local.close()
