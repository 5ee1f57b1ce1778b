"""Multi-GPU: one process per GPU, contiguous shards of independent environments, one collective per step.

The reference has no distributed path at all; this is a new design for the 8-GPU MI355X node.
Environments never interact, so the data path needs no exchange: rank r owns envs [r*n_local, (r+1)*n_local) and
steps them with its own `stg_ctx` (Philox counters use the GLOBAL env index, so results do not depend on the number
of ranks).  The only communication is what an RL learner needs: the step's (obs, reward, terminated, truncated),
54 B/env, packed in one byte buffer and moved by ONE all-gather (RCCL over xGMI when the backend is "nccl"; the
CPU tests run the same code over gloo).  At 131 072 envs per GPU that is 7.1 MB per rank per step.
Actions go the other way: every rank slices its shard out of the global action tensor (a learner that is itself
data-parallel over the same ranks would pass local actions and skip the gather: ``gather=False``).
`gather_begin` / `gather_end` split the collective from the step so that it overlaps the next step's kernel on a
separate HIP stream (a pipelined actor that acts on observations one step old, or a recorder).
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
        nbytes = PACKED_BYTES_PER_ENV * self.n_local
        self._gathered = torch.zeros(self.world * nbytes, dtype=torch.uint8, device=dev)
        # pipelined gather (gather_begin / gather_end): the step's packed outputs are copied to a staging slot and the
        # all-gather runs on its own stream while the next step's kernel runs; two slots, so a slot is reused only
        # after the gather that read it has finished
        self._overlap = dev.type == "cuda" and dist.get_backend(group) != "gloo"
        self._send = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self._recv = [self._gathered, torch.zeros_like(self._gathered)]
        self._slot = 0
        self._pending = None
        if self._overlap:
            self._comm_stream = torch.cuda.Stream(device=dev)
            self._done = [None, None]

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
        return self._unpack_all(self._gathered)

    def _unpack_all(self, buf):
        n = self.n_local
        parts = [unpack_step_buffer(buf[r * PACKED_BYTES_PER_ENV * n:(r + 1) * PACKED_BYTES_PER_ENV * n], n)
                 for r in range(self.world)]
        obs = torch.cat([p[0] for p in parts], dim=1)       # [12, N_global]
        reward = torch.cat([p[1] for p in parts])
        term = torch.cat([p[2] for p in parts])
        trunc = torch.cat([p[3] for p in parts])
        return obs.t(), reward, term.bool(), trunc.bool()

    def gather_begin(self):
        """Starts the all-gather of the step just enqueued and returns at once: the collective runs on a side stream
        (RCCL over xGMI) concurrently with whatever the caller enqueues next -- normally the next step's kernel.  Pair
        with `gather_end`.  One gather may be in flight per slot (two slots)."""
        k = self._slot
        self._slot ^= 1
        packed = self.local.backend.packed
        if not self._overlap:
            self._pending = ("sync", self._gather(unpack=False))
            return
        cur = torch.cuda.current_stream(packed.device)
        if self._done[k] is not None:
            cur.wait_event(self._done[k])                    # the gather that last read this slot has finished
        self._send[k].copy_(packed, non_blocking=True)       # 54 B/env device-to-device, on the compute stream
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ready)
            dist.all_gather_into_tensor(self._recv[k], self._send[k], group=self.group)
            done = torch.cuda.Event()
            done.record(self._comm_stream)
        self._done[k] = done
        self._pending = ("async", k)

    def gather_end(self, unpack: bool = True):
        """Makes the current stream wait for the gather started last and returns its result (see `_gather`)."""
        if self._pending is None:
            raise RuntimeError("gather_end without gather_begin")
        kind, v = self._pending
        self._pending = None
        if kind == "sync":
            buf = v
        else:
            torch.cuda.current_stream(self._recv[v].device).wait_event(self._done[v])
            buf = self._recv[v]
        return self._unpack_all(buf) if unpack else buf

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
