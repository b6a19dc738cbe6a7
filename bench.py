"""bench.py -- env-steps/sec of the batched HIP step engine (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1024 --warmup 128
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole batch: the on-device masked-uniform random
policy picks an action for every env, every env transitions (material flow, sorting, presses,
reward, auto-reset) and obs / mask / reward / done / action are written to HBM for that step.
Workload (config.workload): Env_3_Monolith, 65 536 envs per GPU, max_steps 200, balesize 200,
noise 0.0, env i seeded with its global index (weak scaling: per-GPU work is fixed; no data-path
collective - env instances never interact).  Steps are issued as fused rollout launches of
`--chunk` steps each (default 64; state stays in registers inside a launch); `--chunk 1` gives one
launch per step.  Inputs (state, seeds) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_baseline(kind: str, max_steps: int, noise: float, balesize: int, budget_s: float = 12.0):
    """The CPU oracle (a port of the reference's algorithm, parity-pinned in tests/) on this host's
    cores: same workload shape (masked-uniform random policy, auto-reset), bounded sample."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle.oracle import OracleEnv

    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    # calibrate on one core, then size the sample to ~budget_s of wall time on all cores
    probe = OracleEnv(kind=kind, max_steps=max_steps, seed=0, noise_sorting=noise, balesize=balesize)
    probe.reset(0)
    t0 = time.perf_counter()
    probe.random_rollout(50_000, 2024)
    per_core = 50_000 / (time.perf_counter() - t0)
    steps_each = int(max(50_000, per_core * budget_s))
    envs = [OracleEnv(kind=kind, max_steps=max_steps, seed=i, noise_sorting=noise, balesize=balesize)
            for i in range(cores)]
    for i, e in enumerate(envs):
        e.reset(i)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:  # the C call releases the GIL
        list(ex.map(lambda e: e.random_rollout(steps_each, 2024), envs))
    dt = time.perf_counter() - t0
    total = steps_each * cores
    return {
        "value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{cores} oracle envs x {steps_each} steps ({kind}, masked-uniform policy, auto-reset), "
                  f"{dt:.1f} s wall; single core {per_core:.0f} env-steps/s",
    }


def measured_traffic(args, n, chunk):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01/traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc runs); only quoted for the
    exact workload it was collected on, else null."""
    path = os.path.join(ROOT, "profiles", "r01", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except OSError:
        return None
    if (args.kind, n, chunk, args.noise, args.mode, args.no_outputs) == ("mono", 65536, int(t.get("steps_per_launch", 16)), 0.0, "rollout", False):
        return t.get("hbm_bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--kind", default="mono", choices=["mono", "press", "sort"])
    ap.add_argument("--chunk", type=int, default=64,
                    help="steps fused per rollout launch (the rollout length n_steps of the collector; SB3's PPO default, which the reference trains with, is 2048; DESIGN.md section 6 tabulates 16 / 32 / 64 / 208)")
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--no-outputs", action="store_true", help="diagnostic only: skip obs/mask writes (INVALID as a result)")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"],
                    help="rollout: fused mse_rollout launches (headline); step: mse_sample_actions + mse_step per step")
    ap.add_argument("--sort-policy", default="frozen", choices=["frozen", "rule"],
                    help="Env_2_Pressing only: the sorting decisions - a frozen random policy (per-env Bernoulli(1/2) drawn "
                         "once with seed 1234: BASELINE.json configs[2]) or the reference's sorting_rules()")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2, 3],
                    help="mse_config.rollout_pipeline: 0 by size (default), 1 two-role, 2 one lane per env, 3 three-role ring")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import marl_sortingenv_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    # one rank per GPU over RCCL; with fewer GPUs than ranks (a rehearsal of the N > 1 path on a one-GPU box) the
    # ranks share devices and the two scalar collectives of this script go over gloo
    dev_index = (local_rank % max(1, n_dev)) if world > 1 else 0
    rehearsal = world > 1 and n_dev < world
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", dev_index)

    n = args.envs
    env = M.BatchedSortingEnv(kind=args.kind, num_envs=n, device=dev, base_seed=0, index_offset=rank * n,
                              max_steps=args.max_steps, noise_sorting=args.noise, balesize=200, auto_reset=True,
                              rollout_pipeline=args.pipeline)
    chunk = max(1, args.chunk)
    buf = env.alloc_rollout(chunk, obs=not args.no_outputs, mask=not args.no_outputs)
    stream = torch.cuda.current_stream(dev)
    sort_mode = None
    if args.kind == "press" and args.sort_policy == "frozen":
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        sort_mode = (torch.rand(n, generator=g) < 0.5).to(torch.int32).to(dev)

    def run(n_steps, events=None):
        if args.mode == "step":
            for _ in range(n_steps):
                act = env.sample_actions(policy_seed=2024)
                env.step(act, sort_mode=sort_mode)
            return
        # HIP events on the launch stream bracket the run of launches as a whole: they are queued back to back (the
        # host is far ahead of a 70 us kernel), so elapsed / launches is the kernel's duration plus the ~1.5 us
        # dependent-kernel boundary, not an event round trip per launch.  A last, shorter launch takes the steps
        # that do not fill a chunk (e.g. --steps 5): the roofline divides by the steps actually taken.
        n_full, tail = n_steps // chunk, n_steps % chunk
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        for _ in range(n_full):
            env.rollout(chunk, policy_seed=2024, buffers=buf, sort_mode=sort_mode)
        if tail > 0:
            env.rollout(tail, policy_seed=2024, buffers=buf, sort_mode=sort_mode)
        if events is not None:
            e1.record(stream)
            events.append((n_full + (1 if tail else 0), n_steps, e0, e1))

    def barrier():
        if world > 1:
            dist.barrier()

    run(args.warmup)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    events = []
    t0 = time.perf_counter()
    run(args.steps, events)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # dominant kernel: the rollout kernel.  HIP events on the launch stream around the launches of the timed region
    bytes_per_step = env.algorithmic_bytes_per_step
    if events:
        launches = sum(c for c, _, _, _ in events)
        steps_timed = sum(k for _, k, _, _ in events)
        total_ms = sum(a.elapsed_time(b) for _, _, a, b in events)
        per_launch_ms = total_ms / max(1, launches)
        achieved = bytes_per_step * n * steps_timed / (total_ms * 1e-3) / 1e9
        steps_per_launch_avg = steps_timed / max(1, launches)
    else:  # --mode step: two kernels per step, no single dominant launch to quote
        per_launch_ms, achieved, steps_per_launch_avg = None, None, 1
    assert env.error_count() == 0

    if rank == 0:
        out = {
            "metric": "env-steps/sec (random policy) Env_3_Monolith @65k envs" if args.kind == "mono" and n == 65536
                      else f"env-steps/sec (random policy) {args.kind} @{n} envs",
            "value": n * world * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64 (integer state/masks, fp64 rewards, f32 observations)",
            "data": "synthetic",
            "config": {
                "workload": f"Env_3_Monolith random-policy rollout, {n} envs per GPU" if args.kind == "mono"
                            else f"{args.kind} random-policy rollout, {n} envs per GPU",
                "env_kind": args.kind, **({"sorting_policy": args.sort_policy} if args.kind == "press" else {}),
                "envs_per_gpu": n, "global_envs": n * world, "max_steps": args.max_steps,
                "balesize": 200, "noise_sorting": args.noise, "policy": "on-device masked-uniform, seed 2024",
                "steps_per_launch": chunk, "outputs": "none (diagnostic)" if args.no_outputs else
                "obs f32[K,N,D], mask u8[K,N,A], action i32, reward f32, done u8 per step",
                "parallelism": f"env-index sharding x{world}, no data-path collective" +
                               (f" (REHEARSAL: {world} ranks on {n_dev} GPU)" if rehearsal else ""),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": None if achieved is None else achieved / HBM_PEAK_GBPS,
                "traffic": measured_traffic(args, n, chunk) if args.steps % chunk == 0 else None,
                "kernel": ("k_sample + k_step" if args.mode == "step" else
                           {1: "k_rollout_po", 2: "k_rollout", 3: "k_rollout_ring"}.get(
                               args.pipeline, "k_rollout_ring" if n <= 65536 else "k_rollout")),
                "launch_ms": per_launch_ms,
                "algorithmic_bytes_per_env_step": bytes_per_step, "env_steps_per_launch": n * steps_per_launch_avg,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kind, args.max_steps, args.noise, 200, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
