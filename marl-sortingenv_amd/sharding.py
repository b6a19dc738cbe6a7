"""Env-index sharding over the GPUs of one node: one process per GPU, each with its
own handle; env instances never interact, so stepping needs no collective.  The only exchange is the
rollout-buffer hand-off to the learner: ONE collective per rollout tensor over RCCL / xGMI, issued on a
side stream so that it overlaps the next rollout's stepping - an all-gather (every rank ends up with the whole
rollout: data-parallel learners) or a gather to the learner rank (`mode="gather"`: only rank `dst` allocates the
staging tensors and receives; at 2 M envs the all-gather would deliver 8 x 620 MB per 16-step rollout to every
rank where one rank needs them).

Rank r of R owns global envs [r*n_local, (r+1)*n_local); env g is seeded with base_seed + g and its
policy stream is keyed by g, so the concatenation of all shards equals a single-handle run bit for bit
(tests/test_gpu_batched.py::test_shard_concatenation_equals_single_handle).

Layout of a gathered tensor: RANK-MAJOR [R, K, n_local, ...] - exactly what all_gather_into_tensor writes when
every rank contributes its step-major [K, n_local, ...] buffer as it stands, so the hand-off moves each byte once
(xGMI is point-to-point: ~38.5 MB per rank per step at 262 144 envs) and never transposes.  Global env
g = r * n_local + j sits at [r, :, j]; `step_major()` gives the [K, R, n_local, ...] view (no copy) and
`to_step_major()` the contiguous [K, N, ...] copy for consumers that insist on it.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist


def shard_range(global_envs: int, world_size: int, rank: int):
    """Contiguous block partition; the first (global_envs % world_size) ranks get one env more."""
    base, extra = divmod(int(global_envs), int(world_size))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def step_major(x: torch.Tensor) -> torch.Tensor:
    """[R, K, n, ...] -> the [K, R, n, ...] view of the same memory."""
    return x.transpose(0, 1)


def to_step_major(x: torch.Tensor, sizes=None) -> torch.Tensor:
    """[R, K, n_max, ...] -> contiguous [K, N, ...] (one copy); `sizes`: per-rank env counts of ragged shards."""
    R, K = x.shape[0], x.shape[1]
    if sizes is None:
        return x.transpose(0, 1).reshape((K, R * x.shape[2]) + tuple(x.shape[3:]))
    return torch.cat([x[r, :, : sizes[r]] for r in range(R)], dim=1)


class RolloutExchange:
    """Collective hand-off of step-major rollout tensors [K, n_local, ...] -> rank-major [R, K, n_local, ...].

    Works on any backend (nccl = RCCL on ROCm; gloo in the CPU tests).  Ragged shards are padded to the largest
    one (env counts differ by at most one); `sizes` holds the true counts.

    Overlap without races: `gather_rollout_async(buffers)` enqueues the gather on a side stream behind an event
    recorded on the caller's stream (the rollout that filled `buffers` has finished when the gather reads them)
    and returns a ticket; the caller must not overwrite `buffers` before `ticket.read_done` has passed - rolling
    out into a SECOND buffer set meanwhile (ShardedSortingEnv.rollout_async ping-pongs two sets and waits on that
    event before it reuses one).  The gathered tensors live in two staging sets owned by the exchange, alternated
    per call: a result stays valid until the call after next.  `wait()` makes the caller's stream wait for the gather
    and marks the results as used on it.
    """

    def __init__(self, group=None, device: Optional[torch.device] = None, side_stream: bool = True,
                 force_collective: bool = False, mode: str = "allgather", dst: int = 0):
        if mode not in ("allgather", "gather"):
            raise ValueError("mode must be 'allgather' or 'gather'")
        self.mode, self.dst = mode, int(dst)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.stream = None
        if side_stream and device is not None and device.type == "cuda":
            self.stream = torch.cuda.Stream(device=device)
        self._staging = [dict(), dict()]  # two sets of gathered tensors, keyed by rollout key
        self._consumers = [None, None]    # per staging set: the stream that last consumed its result (wait / release)
        self._turn = 0
        self._pending = None
        self.sizes = None                 # per-rank env counts (set by the first gather)

    # -- helpers --------------------------------------------------------------------------------------------
    def _sizes(self, n_here: int, device) -> list:
        if self.sizes is None:
            if not self.collective:
                self.sizes = [n_here]
            else:
                mine = torch.tensor([n_here], dtype=torch.int64, device=device)
                allv = torch.empty(self.world, dtype=torch.int64, device=device)
                dist.all_gather_into_tensor(allv, mine, group=self.group)
                self.sizes = [int(v) for v in allv.tolist()]
        return self.sizes

    def _gather_into(self, out: torch.Tensor, local: torch.Tensor, n_max: int):
        if local.shape[1] != n_max:  # ragged: pad the env dimension to the largest shard
            padded = torch.zeros((local.shape[0], n_max) + tuple(local.shape[2:]), dtype=local.dtype, device=local.device)
            padded[:, : local.shape[1]] = local
            local = padded
        if not self.collective:
            out[0].copy_(local)
        elif self.mode == "gather":
            # rank-major staging on the learner only: out[r] is rank r's contiguous [K, n_max, ...] block
            dist.gather(local.contiguous(), [out[r] for r in range(out.shape[0])] if out is not None else None,
                        dst=self.dst if self.group is None else dist.get_global_rank(self.group, self.dst), group=self.group)
        else:
            dist.all_gather_into_tensor(out.view((-1,) + tuple(out.shape[2:])), local.contiguous(), group=self.group)

    def _gather_set(self, buffers: Dict[str, Optional[torch.Tensor]], staging: dict) -> dict:
        result = {}
        for key, local in buffers.items():
            if local is None:
                result[key] = None
                continue
            sizes = self._sizes(local.shape[1], local.device)
            n_max = max(sizes)
            shape = (len(sizes), local.shape[0], n_max) + tuple(local.shape[2:])
            if self.collective and self.mode == "gather" and self.rank != self.dst:
                self._gather_into(None, local, n_max)   # a sender: no staging, no result
                result[key] = None
                continue
            out = staging.get(key)
            if out is None or tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device:
                out = staging[key] = torch.empty(shape, dtype=local.dtype, device=local.device)
            self._gather_into(out, local, n_max)
            result[key] = out
        return result

    @property
    def receives(self) -> bool:
        """Does this rank end up with the gathered rollout?  (every rank with 'allgather', rank `dst` with 'gather')"""
        return not (self.collective and self.mode == "gather" and self.rank != self.dst)

    # -- blocking ---------------------------------------------------------------------------------------------
    def gather_rollout(self, buffers: Dict[str, Optional[torch.Tensor]], layout: str = "rank_major"):
        """All-gathers every tensor of a rollout dict (keys as BatchedSortingEnv.alloc_rollout) on the caller's
        stream.  layout "rank_major": [R, K, n_max, ...] staging tensors (valid until the call after next);
        "step_major": contiguous [K, N, ...] copies (the single-handle layout)."""
        self._turn ^= 1
        out = self._gather_set(buffers, self._staging[self._turn])
        if layout == "step_major":
            if not self.receives:
                return {k: None for k in out}
            ragged = len(set(self.sizes)) > 1
            return {k: (None if v is None else to_step_major(v, self.sizes if ragged else None)) for k, v in out.items()}
        return out

    # -- overlapped -------------------------------------------------------------------------------------------
    class Ticket:
        def __init__(self, result, read_done, stream, exchange=None, turn=0):
            self.result, self.read_done, self.stream = result, read_done, stream
            self._exchange, self._turn = exchange, turn

        def release(self, stream=None):
            """Names the stream that consumes `result` (default: the current one).  The staging set is reused by the
            call after next; that gather is then ordered behind everything `stream` has been given up to that moment.
            `wait()` registers the caller's current stream by itself - call release() only when the result is read
            on another stream."""
            if self._exchange is not None and self.stream is not None:
                self._exchange._consumers[self._turn] = stream if stream is not None else \
                    torch.cuda.current_stream(self._exchange.device)

    def gather_rollout_async(self, buffers: Dict[str, Optional[torch.Tensor]]) -> "RolloutExchange.Ticket":
        """Starts the gather of `buffers` on the side stream; see the class docstring for the buffer contract."""
        self._turn ^= 1
        staging = self._staging[self._turn]
        if self.stream is None:  # no device stream (CPU / gloo): blocking
            t = RolloutExchange.Ticket(self._gather_set(buffers, staging), None, None)
            self._pending = t
            return t
        cur = torch.cuda.current_stream(self.device)
        filled = torch.cuda.Event()
        filled.record(cur)                       # the rollout that wrote `buffers`
        self.stream.wait_event(filled)
        consumer = self._consumers[self._turn]
        if consumer is not None:
            # this staging set held the result of the call before last: whatever its consumer's stream has been given
            # so far (its reads of that result, in program order) must finish before the gather overwrites the set
            consumed = torch.cuda.Event()
            consumed.record(consumer)
            self.stream.wait_event(consumed)
            self._consumers[self._turn] = None
        with torch.cuda.stream(self.stream):
            result = self._gather_set(buffers, staging)
            read_done = torch.cuda.Event()
            read_done.record(self.stream)        # `buffers` may be overwritten once this has passed
        for v in buffers.values():
            if v is not None:
                v.record_stream(self.stream)     # the allocator must not recycle them under the gather
        t = RolloutExchange.Ticket(result, read_done, self.stream, self, self._turn)
        self._pending = t
        return t

    def wait(self, ticket: Optional["RolloutExchange.Ticket"] = None):
        """The caller's stream waits for the gather; returns the rank-major result dict."""
        t = ticket if ticket is not None else self._pending
        if t is None:
            return None
        if t.stream is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(t.read_done)
            for v in t.result.values():
                if v is not None:
                    v.record_stream(cur)         # allocated on the side stream, consumed on this one
            t.release(cur)                       # the set's next reuse waits for this stream's reads
        if t is self._pending:
            self._pending = None
        return t.result


class ShardedSortingEnv:
    """This rank's shard of a global batch of envs plus the rollout exchange.

    make_env(n_local, index_offset) -> an object with .rollout(k, policy_seed=..., buffers=...) and
    .alloc_rollout(k); by default a BatchedSortingEnv on this rank's GPU.
    """

    def __init__(self, global_envs: int, kind: str = "mono", base_seed: int = 0, group=None,
                 device: Optional[torch.device] = None, make_env: Optional[Callable] = None,
                 force_collective: bool = False, exchange_mode: str = "allgather", learner_rank: int = 0, **env_kw):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.global_envs = int(global_envs)
        self.start, self.n_local = shard_range(global_envs, self.world, self.rank)
        if make_env is None:
            from .batched import BatchedSortingEnv

            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

            def make_env(n_local, index_offset):
                return BatchedSortingEnv(kind=kind, num_envs=n_local, device=dev, base_seed=base_seed,
                                         index_offset=index_offset, **env_kw)
            device = dev
        self.device = device
        self.env = make_env(self.n_local, self.start)
        self.exchange = RolloutExchange(group=group, device=device, force_collective=force_collective,
                                        mode=exchange_mode, dst=learner_rank)
        self.equal_shards = self.global_envs % self.world == 0
        self._sets = [None, None]    # ping-pong local rollout buffers of rollout_async
        self._tickets = [None, None]
        self._flip = 0

    def rollout(self, k_steps: int, policy_seed: int = 2024, buffers=None, gather: bool = True, layout: str = "step_major"):
        """Blocking: roll out, then gather (default: the single-handle [K, N, ...] layout)."""
        buf = self.env.rollout(k_steps, policy_seed=policy_seed, buffers=buffers)
        if not gather:
            return buf
        return self.exchange.gather_rollout(buf, layout=layout)

    def rollout_async(self, k_steps: int, policy_seed: int = 2024):
        """Rolls out into one of two local buffer sets and starts its gather on the side stream; returns a ticket
        for `collect()`.  The set being gathered is not written again until its gather has read it: before a set is
        reused the stepping stream waits for that set's `read_done` event - two rollouts later, by which time the
        gather has normally long finished, so stepping never stalls on the exchange."""
        i = self._flip
        self._flip ^= 1
        prev = self._tickets[i]
        if prev is not None and prev.read_done is not None:
            torch.cuda.current_stream(self.device).wait_event(prev.read_done)
        if self._sets[i] is None or next(v for v in self._sets[i].values() if v is not None).shape[0] != k_steps:
            alloc = self.env.alloc_rollout(k_steps)
            self._sets[i] = alloc
        buf = self.env.rollout(k_steps, policy_seed=policy_seed, buffers=self._sets[i])
        self._tickets[i] = self.exchange.gather_rollout_async(buf)
        return self._tickets[i]

    def collect(self, ticket):
        """Rank-major gathered rollout of a rollout_async ticket (valid until the second-next rollout_async)."""
        return self.exchange.wait(ticket)
