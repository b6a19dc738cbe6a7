"""Multi-process (world_size 2, gloo, CPU) test of the env-index sharding and the rollout-buffer
hand-off.  The GPU engine cannot run here, so each rank drives a deterministic stand-in whose value
for global env g at step k is a known function of (g, k): the gathered rollout must be exactly the
single-process one, for equal and for ragged shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from marl_sortingenv_amd.sharding import RolloutExchange, ShardedSortingEnv, shard_range, step_major, to_step_major  # noqa: F401


class FakeEnv:
    """rollout buffers as pure functions of (global env index, step)."""

    def __init__(self, n_local, index_offset, obs_dim=5, num_actions=4):
        self.n, self.off, self.D, self.A = n_local, index_offset, obs_dim, num_actions

    def alloc_rollout(self, k):
        return None

    def rollout(self, k, policy_seed=0, buffers=None):
        g = torch.arange(self.off, self.off + self.n, dtype=torch.int64)
        t = torch.arange(k, dtype=torch.int64).unsqueeze(1)
        base = (g.unsqueeze(0) * 1000 + t + policy_seed)
        return {
            "actions": (base % self.A).to(torch.int32),
            "obs": (base.unsqueeze(2) + torch.arange(self.D)).to(torch.float32),
            "reward": (base % 7).to(torch.float32),
            "done": (base % 2).to(torch.uint8),
            "mask": ((base.unsqueeze(2) + torch.arange(self.A)) % 2).to(torch.uint8),
        }


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_envs, k, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = ShardedSortingEnv(global_envs, make_env=lambda n, off: FakeEnv(n, off), device=torch.device("cpu"))
        assert (sh.start, sh.n_local) == shard_range(global_envs, world, rank)
        got = sh.rollout(k, policy_seed=3)
        exp = FakeEnv(global_envs, 0).rollout(k, policy_seed=3)
        for key in exp:
            assert got[key].shape == exp[key].shape, (key, got[key].shape, exp[key].shape)
            assert torch.equal(got[key], exp[key]), key
        # rank-major layout: [R, K, n_max, ...], global env r * n_local + j at [r, :, j]; the step-major view / copy
        rm = sh.rollout(k, policy_seed=3, layout="rank_major")
        sizes = sh.exchange.sizes
        assert sizes == [shard_range(global_envs, world, r)[1] for r in range(world)]
        assert rm["obs"].shape[:3] == (world, k, max(sizes))
        for r in range(world):
            s0 = shard_range(global_envs, world, r)[0]
            assert torch.equal(rm["obs"][r, :, : sizes[r]], exp["obs"][:, s0: s0 + sizes[r]])
        assert step_major(rm["reward"]).shape[:2] == (k, world)
        assert torch.equal(to_step_major(rm["mask"], sizes), exp["mask"])
        # the overlapped API (blocking without a device stream), ping-pong: two tickets alive at once
        fake = FakeEnv(sh.n_local, sh.start)
        ex = sh.exchange
        t1 = ex.gather_rollout_async(fake.rollout(k, policy_seed=5))
        t2 = ex.gather_rollout_async(fake.rollout(k, policy_seed=6))
        for tk, seed in ((t1, 5), (t2, 6)):
            out = ex.wait(tk)
            assert torch.equal(to_step_major(out["obs"], sizes), FakeEnv(global_envs, 0).rollout(k, policy_seed=seed)["obs"])
        # gather-to-learner: only rank `dst` receives (and allocates) the rank-major rollout; the others just send
        for dst in (0, world - 1):
            shg = ShardedSortingEnv(global_envs, make_env=lambda n, off: FakeEnv(n, off), device=torch.device("cpu"),
                                    exchange_mode="gather", learner_rank=dst)
            rm = shg.rollout(k, policy_seed=4, layout="rank_major")
            sm = shg.rollout(k, policy_seed=4)
            exp4 = FakeEnv(global_envs, 0).rollout(k, policy_seed=4)
            assert shg.exchange.receives == (rank == dst)
            if rank == dst:
                for key in exp4:
                    assert torch.equal(to_step_major(rm[key], shg.exchange.sizes), exp4[key]), key
                    assert torch.equal(sm[key], exp4[key]), key
            else:
                assert all(v is None for v in rm.values()) and all(v is None for v in sm.values())
                assert not any(shg.exchange._staging)          # a sender holds no staging memory
            tk = shg.exchange.gather_rollout_async(FakeEnv(shg.n_local, shg.start).rollout(k, policy_seed=8))
            out = shg.exchange.wait(tk)
            if rank == dst:
                assert torch.equal(to_step_major(out["reward"], shg.exchange.sizes), FakeEnv(global_envs, 0).rollout(k, policy_seed=8)["reward"])
        ok[rank] = 1
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("global_envs", [64, 37])
def test_shards_gather_to_single_process_rollout(global_envs):
    world, k = 2, 6
    port = _free_port()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, global_envs, k, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


def test_shard_range_partition():
    for n in (1, 7, 64, 65536, 2097152):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
