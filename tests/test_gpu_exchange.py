"""GPU, one process: the overlapped rollout hand-off (RolloutExchange on its side stream, ping-pong local buffers)
against the blocking path.  world_size 1 with the collective forced on, so the all_gather_into_tensor really runs on
the side stream over RCCL while the next rollout steps into the other buffer set on the main stream."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_side_stream_exchange_equals_blocking_gather():
    import torch
    import torch.distributed as dist

    import marl_sortingenv_amd as M
    from marl_sortingenv_amd.sharding import ShardedSortingEnv, to_step_major

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        n, K, rounds = 8192, 16, 7
        kw = dict(kind="mono", base_seed=3, max_steps=25, noise_sorting=0.05, balesize=200)
        ref = M.BatchedSortingEnv(num_envs=n, device=dev, **kw)
        expect = [{k: v.clone() for k, v in ref.rollout(K, policy_seed=9).items()} for _ in range(rounds)]
        sh = ShardedSortingEnv(n, device=dev, force_collective=True, **kw)
        assert sh.exchange.stream is not None and sh.exchange.collective
        prev, checked = None, 0
        for it in range(rounds):
            ticket = sh.rollout_async(K, policy_seed=9)      # steps into set it % 2 while the gather of it - 1 runs
            if prev is not None:
                got = sh.collect(prev[1])                    # rank-major [1, K, n, ...] staging tensors
                for key, exp in expect[prev[0]].items():
                    assert got[key].shape[:3] == (1, K, n)
                    assert torch.equal(to_step_major(got[key]), exp), (prev[0], key)
                checked += 1
            prev = (it, ticket)
        got = sh.collect(prev[1])
        assert torch.equal(got["obs"][0], expect[rounds - 1]["obs"])
        assert checked == rounds - 1
        # gather-to-learner (dist.gather into the learner's rank-major staging), same overlap, same tensors; the
        # consumer reads on a stream of its own and says so (Ticket.release)
        shg = ShardedSortingEnv(n, device=dev, force_collective=True, exchange_mode="gather", learner_rank=0, **kw)
        assert shg.exchange.receives
        reader = torch.cuda.Stream(device=dev)
        prev, sums = None, []
        for it in range(rounds):
            ticket = shg.rollout_async(K, policy_seed=9)
            if prev is not None:
                got = shg.collect(prev[1])
                reader.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(reader):
                    sums.append((prev[0], got["obs"].double().sum(), got["actions"].long().sum()))
                prev[1].release(reader)
            prev = (it, ticket)
        torch.cuda.synchronize(dev)
        for it, so, sa in sums:
            assert float(so) == float(expect[it]["obs"].double().sum()) and int(sa) == int(expect[it]["actions"].long().sum()), it
        # the blocking path gives the same tensors in the single-handle layout
        ref2 = M.BatchedSortingEnv(num_envs=n, device=dev, **kw)
        sh2 = ShardedSortingEnv(n, device=dev, force_collective=True, **kw)
        a, b = sh2.rollout(K, policy_seed=9), ref2.rollout(K, policy_seed=9)
        for key in b:
            assert torch.equal(a[key], b[key]), key
    finally:
        dist.destroy_process_group()
