"""Multi-GPU: one process per GPU, contiguous shards of independent environments, one collective per step.

The reference has no distributed path at all; this is a new design for the 8-GPU MI355X node.
Environments never interact, so the data path needs no exchange: rank r owns envs [r*n_local, (r+1)*n_local) and steps
them with its own `stg_ctx` (Philox counters use the GLOBAL env index, so results do not depend on the number of ranks).
The only communication is what an RL learner needs -- the step's (obs, reward, terminated, truncated) of every env.

Copy-free by layout.  The step kernel writes its outputs as 56-byte env-major RECORDS (`out_layout='records'`,
include/spintorque_hip.h: STG_OUT_RECORDS) straight into the collective's send buffer -- a shard's records are one
contiguous block -- and ONE all-gather (RCCL over xGMI when the backend is "nccl"; the CPU tests run the same code over
gloo) assembles the GLOBAL record array uint8[N_global, 56]; what the learner gets are typed strided VIEWS of that array --
obs float32 [N_global, 12] (Gym's orientation), reward float32 [N_global], terminated / truncated bool [N_global].  No
staging copy before the collective, no transposition or `torch.cat` after it.  Two send buffers and two global arrays
alternate, so the gather of step k runs on its own HIP stream under the kernel of step k+1 (`gather_begin` /
`gather_end`), which writes the other pair.  At 131 072 envs per GPU a rank contributes 7.3 MB per step.
(`inplace=True` makes the send buffer this rank's slice of the global array itself -- NCCL/RCCL's in-place all-gather
form; the point-to-point exchange always works that way.)

`gather_algo`: "all_gather" (default) = `all_gather_into_tensor`, RCCL picks the algorithm; "p2p" = one grouped batch of
7 sends + 7 receives per rank (`batch_isend_irecv`), every peer's slice over its own xGMI link at once -- the one-shot
alternative to a ring (ring = 7 hops per link-time, direct = 1), for steps short enough that the
collective no longer hides under the kernel; it is also what ragged shards (num_envs % world != 0) use, since it does not
need equal contributions.

Actions go the other way: every rank slices its shard out of the global action tensor (a learner that is itself
data-parallel over the same ranks passes local actions and skips the gather: ``gather=False``).
"""
import time
from typing import Optional

import torch
import torch.distributed as dist

from .backend import RECORD_BYTES, record_views
from .envs import SpinTorqueVecEnv, _TimerTable


def shard_range(n_global: int, world: int, rank: int):
    """Contiguous block partition; the first (n_global % world) ranks take one extra env."""
    base, extra = divmod(n_global, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_views(records: torch.Tensor):
    """(obs [N,12] f32, reward [N] f32, terminated [N] bool, truncated [N] bool): strided views of a record array, no copy."""
    obs, reward, term, trunc, _ = record_views(records)
    return obs, reward, term.view(torch.bool), trunc.view(torch.bool)


class ShardedSpinTorqueVecEnv:
    """`num_envs` global environments sharded over the ranks of `group` (default: the world group)."""

    def __init__(self, num_envs: int, group: Optional[dist.ProcessGroup] = None, device_index: Optional[int] = None,
                 class_index=None, gather_algo: str = "all_gather", inplace: bool = False, overlap: Optional[bool] = None,
                 **env_kwargs):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend 'nccl' = RCCL on ROCm, or 'gloo')")
        if gather_algo not in ("all_gather", "p2p"):
            raise ValueError("gather_algo must be 'all_gather' or 'p2p'")
        self.group = group
        self.gather_algo = gather_algo
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_envs = int(num_envs)
        self.lo, self.hi = shard_range(self.num_envs, self.world, self.rank)
        self.n_local = self.hi - self.lo
        self._spans = [shard_range(self.num_envs, self.world, r) for r in range(self.world)]
        # point-to-point operations address their peer by GLOBAL rank (torch.distributed.P2POp), shards are indexed by the
        # rank inside `group`: the two differ for any group that is not the world
        self._global = [r if group is None else dist.get_global_rank(group, r) for r in range(self.world)]
        if self.num_envs % self.world and gather_algo == "all_gather":
            # ragged shards: an all-gather needs equal contributions; the point-to-point exchange does not care
            gather_algo = self.gather_algo = "p2p"
        if device_index is None:
            device_index = self.rank % max(torch.cuda.device_count(), 1)
        if class_index is not None:
            class_index = torch.as_tensor(class_index)[self.lo:self.hi]
        env_kwargs.pop("out_layout", None)
        self.local = SpinTorqueVecEnv(self.n_local, class_index=class_index, device_index=device_index, env_id0=self.lo,
                                      out_layout="records", **env_kwargs)
        dev = self.local.backend.packed.device
        self.device = dev
        # two global record arrays: the kernel of step k writes this rank's slice of array k & 1 while the gather of
        # step k-1 (array (k-1) & 1) may still be in flight
        self._glob = [torch.zeros((self.num_envs, RECORD_BYTES), dtype=torch.uint8, device=dev) for _ in range(2)]
        # where the step kernel writes: the slice of the global array itself (p2p / in-place all-gather), or a send buffer
        # of its own (the plain out-of-place all-gather, the most travelled path through RCCL)
        self._inplace = bool(inplace) or gather_algo == "p2p"
        self._send = None if self._inplace else [torch.zeros((self.n_local, RECORD_BYTES), dtype=torch.uint8, device=dev)
                                                 for _ in range(2)]
        self._slot = 0            # array the NEXT step writes
        self._filled = None       # array holding the last step's local records, not yet gathered
        self._last = None         # ... whether gathered or not (gather_again)
        self._pending = None
        self._gloo = dist.get_backend(group) == "gloo"
        # the exchange on its own HIP stream, ordered against the step kernels by events.  Automatic: RCCL (stream-ordered
        # collectives).  Over gloo the exchange is a host-side copy out / collective / copy in and gains nothing from a side
        # stream; overlap=True still routes it through the same stream-and-event protocol (the staging copies run on the side
        # stream), which is how the protocol is tested with two processes on ONE GPU, where RCCL cannot run.
        self._overlap = dev.type == "cuda" and (not self._gloo if overlap is None else bool(overlap))
        self._done = [None, None]
        if self._overlap:
            self._comm_stream = torch.cuda.Stream(device=dev)
        self.profiler = _TimerTable()

    # -- the collective ---------------------------------------------------------------------------------------------
    def _mine(self, k: int) -> torch.Tensor:
        """The record array of slot k that this rank's step kernel writes and the exchange sends."""
        return self._glob[k][self.lo:self.hi] if self._inplace else self._send[k]

    def _exchange(self, k: int) -> None:
        """Assembles global array k from every rank's records of slot k."""
        g, mine = self._glob[k], self._mine(k)
        if self._gloo:
            # CPU tests / single-GPU rehearsal: gloo moves host memory and is not an in-place collective
            host = torch.empty(g.shape, dtype=torch.uint8)
            # (device tensors: .cpu() / copy_ run on the CURRENT stream -- the side stream when the caller is gather_begin with
            # overlap -- and block the host until that stream has reached them)
            if self.gather_algo == "p2p" and self.world > 1:
                src = mine.cpu()
                ops = []
                for peer, (plo, phi) in enumerate(self._spans):
                    if peer != self.rank:
                        ops.append(dist.P2POp(dist.isend, src, self._global[peer], group=self.group))
                        ops.append(dist.P2POp(dist.irecv, host[plo:phi], self._global[peer], group=self.group))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                host[self.lo:self.hi] = src
            else:
                dist.all_gather_into_tensor(host.view(-1), mine.cpu().reshape(-1), group=self.group)
            g.copy_(host)
            return
        if self.gather_algo == "p2p" and self.world > 1:
            ops = []
            for peer, (plo, phi) in enumerate(self._spans):
                if peer != self.rank:
                    ops.append(dist.P2POp(dist.isend, mine, self._global[peer], group=self.group))
                    ops.append(dist.P2POp(dist.irecv, g[plo:phi], self._global[peer], group=self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()            # (stream-ordered on NCCL/RCCL: does not block the host)
            return
        # (in-place form: the send buffer is this rank's slice of the receive buffer, NCCL/RCCL's in-place all-gather)
        dist.all_gather_into_tensor(g.view(-1), mine.reshape(-1), group=self.group)

    def _gather(self, unpack: bool = True):
        """Synchronous form: gathers the last step's records and returns the typed global views (unpack=False: the
        global record array uint8 [N_global, 56] itself)."""
        self.gather_begin()
        return self.gather_end(unpack)

    def gather_begin(self):
        """Starts the all-gather of the step just enqueued and returns at once: the collective runs on a side stream
        (RCCL over xGMI) concurrently with whatever the caller enqueues next -- normally the next step's kernel, which
        writes the OTHER global array.  Pair with `gather_end`."""
        if self._filled is None:
            raise RuntimeError("gather_begin without a step (or reset) to gather")
        k, self._filled = self._filled, None
        t0 = time.perf_counter()
        if not self._overlap:
            self._exchange(k)
            self._pending = ("sync", k)
        else:
            cur = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(cur)                                    # the step kernel that wrote this rank's slice
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                self._exchange(k)
                done = torch.cuda.Event()
                done.record(self._comm_stream)
            self._done[k] = done
            self._pending = ("async", k)
        self.profiler.add("gather_begin", time.perf_counter() - t0)

    def gather_end(self, unpack: bool = True):
        """Makes the current stream wait for the gather started last; returns (obs [N,12], reward [N], terminated [N],
        truncated [N]) as views of the global record array (valid until the step after next overwrites that array), or
        the array itself with unpack=False."""
        if self._pending is None:
            raise RuntimeError("gather_end without gather_begin")
        kind, k = self._pending
        self._pending = None
        if kind == "async":
            torch.cuda.current_stream(self.device).wait_event(self._done[k])
        return global_views(self._glob[k]) if unpack else self._glob[k]

    def gather_again(self) -> None:
        """Marks the records of the last step as not yet gathered, so that gather_begin() exchanges them once more
        (measuring the collective by itself)."""
        if self._last is None:
            raise RuntimeError("nothing has been stepped yet")
        self._filled = self._last

    @property
    def gather_in_flight(self) -> bool:
        """A gather_begin() is waiting for its gather_end()."""
        return self._pending is not None

    # -- env API ----------------------------------------------------------------------------------------------------
    def _next_slot(self) -> int:
        k = self._slot
        self._slot ^= 1
        if self._overlap and self._done[k] is not None:
            # the gather that last filled this array must be through before the kernel overwrites our slice of it
            torch.cuda.current_stream(self.device).wait_event(self._done[k])
        return k

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
        # reset() fills the env's own record array (obs fields; reward/flags zeroed): hand it to the exchange
        k = self._next_slot()
        self._mine(k).copy_(self.local.backend.packed)
        self._filled = self._last = k
        return self._gather()[0], info

    def step(self, actions, gather: bool = True, actions_are_local: bool = False, actions_soa: bool = False):
        """One env.step() of this rank's shard, written into its slice of the next global record array.  gather=True
        also runs the exchange and returns the GLOBAL (obs, reward, terminated, truncated, {}); gather=False returns the
        local env's outputs (views of this rank's slice) -- follow with gather_begin()/gather_end() to pipeline."""
        a = torch.as_tensor(actions)
        if not actions_are_local:
            a = a[:, self.lo:self.hi] if actions_soa else a[self.lo:self.hi]
        k = self._next_slot()
        out = self.local.step(a, actions_soa=actions_soa, out=self._mine(k))
        self._filled = self._last = k
        if not gather:
            return out
        obs, reward, term, trunc = self._gather()
        return obs, reward, term, trunc, {}

    def scatter_actions(self, actions=None, src: int = 0, dtype=torch.float32):
        """A centralised learner's actions -> this rank's shard: `actions` [N_global, 2] on rank `src` (ignored elsewhere),
        8 B/env over the same links as the gather, in the other direction.  Returns the local [n_local, 2] tensor; pass it
        to step(..., actions_are_local=True)."""
        dev = torch.device("cpu") if self._gloo else self.device
        local = torch.empty((self.n_local, 2), dtype=dtype, device=dev)
        chunks = None
        if self.rank == src:
            a = torch.as_tensor(actions).to(device=dev, dtype=dtype)
            if tuple(a.shape) != (self.num_envs, 2):
                raise ValueError(f"expected actions of shape ({self.num_envs}, 2), got {tuple(a.shape)}")
            chunks = [a[plo:phi].contiguous() for plo, phi in self._spans]
        gsrc = self._global[src]
        if self.num_envs % self.world == 0:
            dist.scatter(local, chunks, src=gsrc, group=self.group)
            return local
        # ragged shards: one send per peer (scatter needs equal chunks)
        if self.rank == src:
            ops = [dist.P2POp(dist.isend, chunks[peer], self._global[peer], group=self.group) for peer in range(self.world) if peer != src]
            local.copy_(chunks[src])
        else:
            ops = [dist.P2POp(dist.irecv, local, gsrc, group=self.group)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return local

    def get_performance_stats(self):
        st = self.local.get_performance_stats()
        st["profiler"].update(self.profiler.get_stats())
        return st

    def close(self):
        self.local.close()
