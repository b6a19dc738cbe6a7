"""End-to-end rehearsal of the N > 1 path on a ONE-GPU box: R ranks share the device, gloo carries the exchange.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 \\
           tools/rehearse_sharding.py

Every rank steps its shard of a ragged global batch with the HIP engine (ShardedSortingEnv, env-index sharding),
the rollout buffers are all-gathered (the hand-off that is one RCCL all_gather_into_tensor per rollout on a real
multi-GPU node; here gloo on host copies), and rank 0 compares the gathered rollout with a single-handle run of the
whole batch: bit for bit, over several rollouts.  On a multi-GPU node the same script runs with one GPU per rank over
nccl (it picks the backend from the device count).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import marl_sortingenv_amd as M  # noqa: E402
from marl_sortingenv_amd.sharding import RolloutExchange, shard_range  # noqa: E402

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
local_rank = int(os.environ.get("LOCAL_RANK", rank))
n_dev = torch.cuda.device_count()
shared = n_dev < world
dev = torch.device("cuda", local_rank % n_dev)
torch.cuda.set_device(dev)
dist.init_process_group(backend="gloo" if shared else "nccl")

G, K = 10007, 20  # ragged: 10007 envs do not divide by 2, 3 or 4
start, n_local = shard_range(G, world, rank)
kw = dict(kind="mono", base_seed=42, max_steps=15, noise_sorting=0.05, balesize=200)
env = M.BatchedSortingEnv(num_envs=n_local, device=dev, index_offset=start, **kw)
ex = RolloutExchange(device=torch.device("cpu") if shared else dev, side_stream=not shared)
full = M.BatchedSortingEnv(num_envs=G, device=dev, **kw) if rank == 0 else None
ok = True
for it in range(3):
    buf = env.rollout(K, policy_seed=9)
    if shared:  # gloo: exchange host copies
        buf = {k: v.cpu() for k, v in buf.items()}
    got = ex.gather_rollout(buf, layout="step_major")
    if rank == 0:
        ref = full.rollout(K, policy_seed=9)
        for k in ref:
            same = torch.equal(got[k].to(dev), ref[k])
            ok = ok and same
            if not same:
                print(f"rollout {it}: {k} differs", flush=True)
if rank == 0:
    print(f"sharding rehearsal {'OK' if ok else 'FAILED'}: {world} ranks x ~{G // world} envs on {n_dev} GPU(s), "
          f"{'gloo (shared device)' if shared else 'nccl'}, 3 rollouts of {K} steps identical to one handle of {G} envs",
          flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
