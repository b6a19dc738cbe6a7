"""Env-index sharding over the GPUs of one node: one process per GPU, each with its
own handle; env instances never interact, so stepping needs no collective.  The only exchange is the
rollout-buffer hand-off to the learner: an all-gather (or gather to one rank) over RCCL / xGMI, issued
once per rollout on a side stream so it overlaps the next rollout's stepping.

Rank r of R owns global envs [r*n_local, (r+1)*n_local); env g is seeded with base_seed + g and its
policy stream is keyed by g, so the concatenation of all shards equals a single-handle run bit for bit
(tests/test_gpu_batched.py::test_shard_concatenation_equals_single_handle).
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


class RolloutExchange:
    """Collective hand-off of step-major rollout tensors [K, n_local, ...] -> [K, n_global, ...].

    Works on any backend (nccl = RCCL on ROCm; gloo in the CPU tests).  Shards must be equal-sized
    for the single all_gather_into_tensor; ragged shards are padded to the largest one first.
    """

    def __init__(self, group=None, device: Optional[torch.device] = None, side_stream: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device
        self.stream = None
        if side_stream and device is not None and device.type == "cuda":
            self.stream = torch.cuda.Stream(device=device)
        self._pending = None

    def all_gather(self, local: torch.Tensor, env_dim: int = 1, equal_shards: bool = True) -> torch.Tensor:
        """Blocking gather of one tensor along its env dimension."""
        if self.world == 1:
            return local
        x = local.movedim(env_dim, 0).contiguous()  # [n_local, ...]: env-major so that shards concatenate
        if equal_shards:
            out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(out, x, group=self.group)
        else:
            # ragged shards: pad to the largest shard, gather, drop the padding
            n_here = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
            sizes = torch.empty(self.world, dtype=torch.int64, device=x.device)
            dist.all_gather_into_tensor(sizes, n_here, group=self.group)
            sizes = [int(v) for v in sizes.tolist()]
            n_max = max(sizes)
            padded = torch.zeros((n_max,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            padded[: x.shape[0]] = x
            allp = torch.empty((self.world * n_max,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(allp, padded, group=self.group)
            out = torch.cat([allp[r * n_max: r * n_max + sizes[r]] for r in range(self.world)], dim=0)
        return out.movedim(0, env_dim)

    def gather_rollout(self, buffers: Dict[str, Optional[torch.Tensor]], equal_shards: bool = True):
        """All-gathers every tensor of a rollout dict (keys as BatchedSortingEnv.alloc_rollout)."""
        return {k: (None if v is None else self.all_gather(v, env_dim=1, equal_shards=equal_shards))
                for k, v in buffers.items()}

    # -- overlapped variant: launch on the side stream, collect later -----------------------------
    def gather_rollout_async(self, buffers: Dict[str, Optional[torch.Tensor]]):
        if self.stream is None:
            self._pending = self.gather_rollout(buffers)
            return
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)  # the rollout kernel that filled `buffers` must have finished
        with torch.cuda.stream(self.stream):
            self._pending = self.gather_rollout(buffers)

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        out, self._pending = self._pending, None
        return out


class ShardedSortingEnv:
    """This rank's shard of a global batch of envs plus the rollout exchange.

    make_env(n_local, index_offset) -> an object with .rollout(k, policy_seed=..., buffers=...) and
    .alloc_rollout(k); by default a BatchedSortingEnv on this rank's GPU.
    """

    def __init__(self, global_envs: int, kind: str = "mono", base_seed: int = 0, group=None,
                 device: Optional[torch.device] = None, make_env: Optional[Callable] = None, **env_kw):
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
        self.env = make_env(self.n_local, self.start)
        self.exchange = RolloutExchange(group=group, device=device)
        self.equal_shards = self.global_envs % self.world == 0

    def rollout(self, k_steps: int, policy_seed: int = 2024, buffers=None, gather: bool = True):
        buf = self.env.rollout(k_steps, policy_seed=policy_seed, buffers=buffers)
        if not gather:
            return buf
        return self.exchange.gather_rollout(buf, equal_shards=self.equal_shards)
