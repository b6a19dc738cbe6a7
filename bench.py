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

A pass = exactly `--steps` steps.  One pass of a short run (the driver's `--steps 20` is one 85 us launch) is
shorter than the two device synchronisations that bracket it, so the timed region holds `reps` passes issued back
to back (state advancing, `reps` reported; default: as many as reach ~65 536 steps = 0.2 s, the length from
which the per-launch time stops falling - a region of a few ms is spent in the GPU's clock ramp; `--reps 1` for a
single pass) and value / ms_per_step are per step of that region; `single_pass_ms` is the median of individually
bracketed passes, sync overhead included, for comparison.

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
# SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / waves per SIMD), profiles/r02/summary.json (tags sq_k64, sq_mlp_262144): the
# share of cycles in which a SIMD's VALU is issuing - k_rollout_ring 0.85 (3 waves per SIMD), k_rollout_policy 0.88 (2)
VALU_ISSUE_RATIO = {"random": 0.85, "mlp": 0.88}
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak


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


def measured_traffic(args, n, steps_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r02/traffic.json,
    else r01's: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc runs, tools/prof_traffic.sh); only quoted for the
    exact workload and steps per launch it was collected on, else null."""
    if (args.kind, args.noise, args.mode, args.no_outputs) != ("mono", 0.0, "rollout", False):
        return None
    for rnd in ("r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "traffic.json")) as f:
                t = json.load(f)
        except OSError:
            continue
        entries = t.get("entries") or [t]
        for e in entries:
            if int(e.get("steps_per_launch", -1)) == int(steps_per_launch) and e.get("policy", "random") == args.policy \
                    and int(e.get("envs", 65536)) == n:
                return e.get("hbm_bytes_per_launch")
    return None


# What one launch of K fused steps must move per env (SURVEY 8d's inventory): every step's
# outputs and action, the state once per launch.  {kind: (state read, state written, outputs + action per step)}
FUSED_BYTES = {"mono": (130, 114, 143 + 4), "sort": (130 + 37, 114 + 21, 52 + 4 + 1 + 4), "press": (130, 114, 64 + 11 + 4 + 1 + 4)}


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
    ap.add_argument("--reps", type=int, default=0,
                    help="passes of --steps steps inside the timed region (0 = as many as reach ~65 536 steps)")
    ap.add_argument("--policy", default="random", choices=["random", "mlp"],
                    help="random: the on-device masked-uniform policy (headline); mlp: the reference's actor-critic MLP "
                         "evaluated inside the rollout kernel (MaskablePPO-shaped collection, BASELINE.json configs[3])")
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "f16x3"],
                    help="--policy mlp: arithmetic of the policy's matrix products (include/mse.h mse_policy_set_precision)")
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

    collector = None
    if args.policy == "mlp":
        collector = M.FusedPolicyRollout(env, M.MlpPolicy.random_init(env.obs_dim, env.num_actions, seed=7, device=dev, precision=args.precision), chunk)

    def run(n_steps, events=None):
        if collector is not None:
            n_full, tail = n_steps // chunk, n_steps % chunk
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            for _ in range(n_full):
                collector.collect(chunk)
            if tail > 0:
                collector.collect(tail)
            if events is not None:
                e1.record(stream)
                events.append((n_full + (1 if tail else 0), n_steps, e0, e1))
            return
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

    # default: ~65 536 steps (0.2 s) in the timed region.  Shorter regions measure the clock ramp, not the kernel: at
    # --steps 20 a 8 ms region gives 75-77 us per launch, 70 ms 69.7, 0.27 s 68.6, 1.1 s 68.3
    reps = args.reps if args.reps > 0 else max(1, -(-65536 // args.steps))
    run(args.warmup)
    torch.cuda.synchronize(dev)
    # individually bracketed passes first (sync on both sides of each: what a caller who waits per pass sees)
    singles = []
    for _ in range(min(reps, 9)):
        t1 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize(dev)
        singles.append(time.perf_counter() - t1)
    singles.sort()
    barrier()
    torch.cuda.synchronize(dev)
    # one HIP event before the first launch of the region and one after the last (an event pair per pass would put two
    # event records between consecutive launches: ~10 us per 80 us launch at --steps 20)
    events = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per_pass = args.steps // chunk + (1 if args.steps % chunk else 0)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(reps):
        run(args.steps)
    ev1.record(stream)
    if args.mode != "step":
        events.append((per_pass * reps, args.steps * reps, ev0, ev1))
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # dominant kernel: the rollout kernel.  HIP events on the launch stream around the launches of the timed region
    # (first event of the first pass to the last event of the last: the passes are queued back to back)
    bytes_contract = env.algorithmic_bytes_per_step  # charges a state round trip to EVERY step
    if events:
        launches = sum(c for c, _, _, _ in events)
        steps_timed = sum(k for _, k, _, _ in events)
        total_ms = events[0][2].elapsed_time(events[-1][3])
        per_launch_ms = total_ms / max(1, launches)
        steps_per_launch_avg = steps_timed / max(1, launches)
        st_r, st_w, per_step = FUSED_BYTES[args.kind]
        noise_r, noise_w = (32, 16) if args.noise > 0 else (0, 0)
        # bytes a fused launch must move per env-step: outputs + action every step, state once per launch
        policy_out = 8 if args.policy == "mlp" else 0  # + log-probability and value f32 per step
        fused_bytes = per_step + policy_out + (st_r + st_w + noise_r + noise_w) / steps_per_launch_avg
        achieved = fused_bytes * n * steps_timed / (total_ms * 1e-3) / 1e9
        achieved_contract = bytes_contract * n * steps_timed / (total_ms * 1e-3) / 1e9
    else:  # --mode step: two kernels per step, no single dominant launch to quote
        per_launch_ms, achieved, achieved_contract, steps_per_launch_avg, fused_bytes = None, None, None, 1, None
    assert env.error_count() == 0

    if rank == 0:
        out = {
            "metric": "env-steps/sec (random policy) Env_3_Monolith @65k envs" if args.kind == "mono" and n == 65536
                      else f"env-steps/sec (random policy) {args.kind} @{n} envs",
            "value": n * world * args.steps * reps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "reps": reps,
            "ms_per_step": elapsed * 1e3 / (args.steps * reps),
            "single_pass_ms": singles[len(singles) // 2] * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64 (integer state/masks, fp64 rewards, f32 observations)",
            "data": "synthetic",
            "config": {
                "workload": f"Env_3_Monolith random-policy rollout, {n} envs per GPU" if args.kind == "mono"
                            else f"{args.kind} random-policy rollout, {n} envs per GPU",
                "env_kind": args.kind, **({"sorting_policy": args.sort_policy} if args.kind == "press" else {}),
                "envs_per_gpu": n, "global_envs": n * world, "max_steps": args.max_steps,
                "balesize": 200, "noise_sorting": args.noise,
                "policy": "on-device masked-uniform, seed 2024" if args.policy == "random" else
                          f"actor-critic MLP 2x32 tanh (random-init weights, matrix products {collector.policy.precision}) with masked categorical sampling, inside the rollout kernel",
                "steps_per_launch": min(chunk, args.steps), "outputs": "none (diagnostic)" if args.no_outputs else
                "obs f32[K,N,D], mask u8[K,N,A], action i32, reward f32, done u8 per step",
                "parallelism": f"env-index sharding x{world}, no data-path collective" +
                               (f" (REHEARSAL: {world} ranks on {n_dev} GPU)" if rehearsal else ""),
            },
            "roofline": {
                # priced against HBM (the contract's roofline for this byte/integer path); what actually limits the
                # kernel at this size is VALU issue: SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / waves per SIMD) from the
                # committed PMC pass (profiles/)
                "bound": "valu-issue", "priced_against": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": None if achieved is None else achieved / HBM_PEAK_GBPS,
                "fused_bytes_per_env_step": fused_bytes,
                "traffic": measured_traffic(args, n, steps_per_launch_avg) if float(steps_per_launch_avg).is_integer() else None,
                "kernel": ("k_sample + k_step" if args.mode == "step" else
                           "k_rollout_policy" if args.policy == "mlp" else
                           {1: "k_rollout_po", 2: "k_rollout", 3: "k_rollout_ring"}.get(
                               args.pipeline, "k_rollout_ring" if n <= 65536 else "k_rollout")),
                "launch_ms": per_launch_ms, "env_steps_per_launch": n * steps_per_launch_avg,
                # SURVEY 8d's per-step contract (state read + written EVERY step): a fused launch does not move those
                # bytes, so this figure can pass 1.0 on long launches of large batches - kept for comparison only
                "contract_bytes_per_env_step": bytes_contract, "achieved_contract": achieved_contract,
                "frac_contract": None if achieved_contract is None else achieved_contract / HBM_PEAK_GBPS,
                "valu_issue_ratio": VALU_ISSUE_RATIO[args.policy],
            },
        }
        if collector is not None and per_launch_ms:
            # the policy network's share: 10 240 flop per env and forward (5 layers of 32 x 32 x 2), one forward per
            # step plus the critic's two layers (4 096 flop) for the bootstrap value per launch; in the f16x3 form each
            # product costs three f16 MFMAs
            f32_equiv = n * (10240.0 * steps_per_launch_avg + 4096.0) / (per_launch_ms * 1e-3) / 1e12
            split = collector.policy.precision == "f16x3"
            out["roofline"]["mfma"] = {
                "form": collector.policy.precision, "f32_equivalent_tflops": f32_equiv,
                "achieved": f32_equiv * (3.0 if split else 1.0), "peak": F16_MFMA_PEAK_TFLOPS if split else 157.3,
                "unit": "TFLOP/s", "frac": f32_equiv * (3.0 if split else 1.0) / (F16_MFMA_PEAK_TFLOPS if split else 157.3),
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kind, args.max_steps, args.noise, 200, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
