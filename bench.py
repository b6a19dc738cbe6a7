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

A pass = exactly `--steps` steps.  One pass of a short run (the driver's `--steps 20` is one ~65 us launch) is
shorter than the two device synchronisations that bracket it, so the timed region holds `reps` passes issued back
to back (state advancing; default: as many as reach ~65 536 steps = 0.2 s, the length from which the per-launch time
stops falling - a region of a few ms is spent in the GPU's clock ramp; `--reps 1` for a single pass).  `value` and
`ms_per_step` are per step of that region, whose length the line states: `timed_steps` = steps x reps and
`timed_region_s`.  `single_pass_ms` / `single_pass_value` are what ONE individually bracketed pass gives (median of 9,
sync overhead included).

One GPU: the line also carries `configs_measured` - short sub-runs of BASELINE.json's other single-GPU configurations
(configs[1..3]: Env_1 / Env_2 at 65 536 envs, Env_3 at 262 144 envs with the random and the learned policy in both
product forms, the class-default noise) on the same clock, each with its kernel, launch time and roofline fraction.
N GPUs: after the collective-free headline region a second region runs the rollout hand-off (BASELINE.json configs[4],
`--exchange allgather|gather`: RCCL on a side stream, overlapped with the next rollout) and reports it under `exchange`.

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
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak
F32_MFMA_PEAK_TFLOPS = 157.3

# What one launch of K fused steps must move per env (SURVEY 8d's inventory): every step's
# outputs and action, the state once per launch.  {kind: (state read, state written, outputs + action per step)}
FUSED_BYTES = {"mono": (130, 114, 143 + 4), "sort": (130 + 37, 114 + 21, 52 + 4 + 1 + 4), "press": (130, 114, 64 + 11 + 4 + 1 + 4)}
KIND_NAME = {"mono": "Env_3_Monolith", "press": "Env_2_Pressing", "sort": "Env_1_Sorting"}


def cpu_baseline(kind: str, max_steps: int, noise: float, balesize: int, budget_s: float = 8.0):
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
    out = {
        "value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{cores} oracle envs x {steps_each} steps ({kind}, masked-uniform policy, auto-reset), "
                  f"{dt:.1f} s wall; single core {per_core:.0f} env-steps/s",
    }
    ref = reference_numpy()
    if ref is not None:
        out["reference_numpy"] = ref
    return out


def reference_numpy():
    """The reference's own NumPy env, timed in the BUILD CONTAINER by tools/time_reference.py (it cannot travel to the
    GPU box and is never run from here): quoted from the committed file, with its provenance."""
    for rnd in ("r03",):
        path = os.path.join(ROOT, "profiles", rnd, "reference_cpu.json")
        try:
            with open(path) as f:
                d = json.load(f)
        except OSError:
            continue
        return {
            "provenance": f"profiles/{rnd}/reference_cpu.json, written by tools/time_reference.py in the build container "
                          f"({d.get('cpu_model')}, {d.get('cores_used')} of {d.get('cpu_count')} cores, numpy {d.get('numpy')}); "
                          "NOT measured on this box",
            "unit": "env-steps/s",
            "configs": {k: v.get("env_steps_per_s") for k, v in d.get("configs", {}).items()},
        }
    return None


def profiled(kind, n, noise, policy, steps_per_launch, key):
    """A figure from the committed rocprofv3 PMC passes (profiles/rNN/traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    separate --pmc runs; `valu_issue_ratio` = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / waves per SIMD)): only quoted for
    the exact workload and steps per launch it was collected on, else null."""
    if not float(steps_per_launch).is_integer():
        return None
    for rnd in ("r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "traffic.json")) as f:
                t = json.load(f)
        except OSError:
            continue
        for e in t.get("entries") or [t]:
            if int(e.get("steps_per_launch", -1)) == int(steps_per_launch) and e.get("policy", "random") == policy \
                    and int(e.get("envs", 65536)) == n and e.get("env_kind", "mono") == kind \
                    and float(e.get("noise", 0.0)) == float(noise) and e.get(key) is not None:
                return e.get(key)
    return None


def kernel_name(kind, n, policy, pipeline, mode, cus=256, precision="f16x3"):
    if mode == "step":
        return "k_sample + k_step"
    if policy == "mlp":  # the library's rule (launch_rollout_policy): roles while a SIMD gets 64 envs, f16x3 form only
        return "k_rollout_policy_roles" if n <= 256 * cus and pipeline != 2 and precision != "f32" else "k_rollout_policy"
    wgs = -(-n // 256)   # the library's rule (mse_create): one round of 256-env workgroups, or most of a second one
    by_size = wgs <= cus or (cus + cus // 4 < wgs <= 2 * cus)
    return {1: "k_rollout_po", 2: "k_rollout", 3: "k_rollout_ring"}.get(pipeline, "k_rollout_ring" if by_size else "k_rollout")


class Workload:
    """One configuration on one GPU: its handle, its buffers, and `run(n_steps)` = that many steps issued as fused
    launches of `chunk` steps (a last, shorter launch takes what does not fill a chunk)."""

    def __init__(self, M, torch, dev, *, kind, n, chunk, noise=0.0, max_steps=200, policy="random", precision="auto",
                 pipeline=0, sort_policy="frozen", mode="rollout", no_outputs=False, index_offset=0, rank=0):
        self.torch, self.dev, self.kind, self.n, self.chunk, self.noise = torch, dev, kind, n, max(1, chunk), noise
        self.policy, self.mode, self.pipeline, self.no_outputs, self.max_steps = policy, mode, pipeline, no_outputs, max_steps
        self.sort_policy = sort_policy
        self.env = M.BatchedSortingEnv(kind=kind, num_envs=n, device=dev, base_seed=0, index_offset=index_offset,
                                       max_steps=max_steps, noise_sorting=noise, balesize=200, auto_reset=True,
                                       rollout_pipeline=pipeline)
        self.buf = self.env.alloc_rollout(self.chunk, obs=not no_outputs, mask=not no_outputs)
        self.stream = torch.cuda.current_stream(dev)
        self.sort_mode = None
        if kind == "press" and sort_policy == "frozen":
            g = torch.Generator(device="cpu").manual_seed(1234 + rank)
            self.sort_mode = (torch.rand(n, generator=g) < 0.5).to(torch.int32).to(dev)
        self.collector = None
        if policy == "mlp":
            self.collector = M.FusedPolicyRollout(
                self.env, M.MlpPolicy.random_init(self.env.obs_dim, self.env.num_actions, seed=7, device=dev, precision=precision), self.chunk)

    def launches(self, n_steps):
        return n_steps // self.chunk + (1 if n_steps % self.chunk else 0)

    def run(self, n_steps):
        chunk = self.chunk
        if self.mode == "step":
            for _ in range(n_steps):
                self.env.step(self.env.sample_actions(policy_seed=2024), sort_mode=self.sort_mode)
            return
        n_full, tail = n_steps // chunk, n_steps % chunk
        if self.collector is not None:
            for _ in range(n_full):
                self.collector.collect(chunk)
            if tail > 0:
                self.collector.collect(tail)
            return
        for _ in range(n_full):
            self.env.rollout(chunk, policy_seed=2024, buffers=self.buf, sort_mode=self.sort_mode)
        if tail > 0:
            self.env.rollout(tail, policy_seed=2024, buffers=self.buf, sort_mode=self.sort_mode)

    def fused_bytes(self, steps_per_launch):
        """bytes a fused launch must move per env-step: outputs + action every step, state once per launch"""
        st_r, st_w, per_step = FUSED_BYTES[self.kind]
        noise_r, noise_w = (32, 16) if self.noise > 0 else (0, 0)
        policy_out = 8 if self.policy == "mlp" else 0  # + log-probability and value f32 per step
        return per_step + policy_out + (st_r + st_w + noise_r + noise_w) / steps_per_launch

    def describe(self):
        pol = "masked-uniform random policy" if self.policy == "random" else \
            f"learned policy (actor-critic MLP 2x32, {self.collector.policy.precision} products) in the rollout kernel"
        extra = f", sorting decisions: {self.sort_policy}" if self.kind == "press" else ""
        noise = f", noise_sorting {self.noise}" if self.noise > 0 else ""
        return f"{KIND_NAME[self.kind]} {pol}, {self.n} envs per GPU{extra}{noise}"

    def close(self):
        self.env.close()
        self.buf = self.collector = None


def timed_region(torch, dev, wl, steps, reps, barrier=lambda: None, singles=9):
    """`reps` passes of `steps` steps issued back to back, bracketed by barrier + synchronize on both sides (wall clock)
    and by ONE pair of HIP events on the launch stream (an event pair per pass would put two event records between
    consecutive launches: ~10 us per 65 us launch at --steps 20).  Returns wall seconds, event ms, and the median of
    individually bracketed single passes."""
    single = []
    for _ in range(min(reps, singles)):
        t1 = time.perf_counter()
        wl.run(steps)
        torch.cuda.synchronize(dev)
        single.append(time.perf_counter() - t1)
    single.sort()
    barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(wl.stream)
    for _ in range(reps):
        wl.run(steps)
    ev1.record(wl.stream)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    return elapsed, ev0.elapsed_time(ev1), (single[len(single) // 2] if single else None)


def roofline_of(wl, steps, reps, event_ms):
    """The dominant kernel's figures from the HIP events around the timed region's launches (queued back to back: the
    host is far ahead of a 65 us kernel, so elapsed / launches is the kernel's duration plus the ~1.5 us
    dependent-kernel boundary)."""
    launches = wl.launches(steps) * reps
    steps_timed = steps * reps
    per_launch_ms = event_ms / max(1, launches)
    spl = steps_timed / max(1, launches)
    fused = wl.fused_bytes(spl)
    achieved = fused * wl.n * steps_timed / (event_ms * 1e-3) / 1e9
    contract = wl.env.algorithmic_bytes_per_step  # charges a state round trip to EVERY step
    achieved_contract = contract * wl.n * steps_timed / (event_ms * 1e-3) / 1e9
    r = {
        # priced against HBM (the contract's roofline for this byte/integer path); what actually limits the kernel at
        # this size is VALU issue (valu_issue_ratio, from the committed PMC pass of this exact configuration, else null)
        "bound": "valu-issue", "priced_against": "hbm",
        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "fused_bytes_per_env_step": fused,
        "traffic": None if wl.no_outputs else profiled(wl.kind, wl.n, wl.noise, wl.policy, spl, "hbm_bytes_per_launch"),
        "kernel": kernel_name(wl.kind, wl.n, wl.policy, wl.pipeline, wl.mode,
                              precision=wl.collector.policy.precision if wl.policy == "mlp" else "f16x3"),
        "launch_ms": per_launch_ms, "env_steps_per_launch": wl.n * spl,
        # SURVEY 8d's per-step contract (state read + written EVERY step): a fused launch does not move those bytes,
        # so this figure can pass 1.0 on long launches of large batches - kept for comparison only
        "contract_bytes_per_env_step": contract, "achieved_contract": achieved_contract,
        "frac_contract": achieved_contract / HBM_PEAK_GBPS,
        "valu_issue_ratio": profiled(wl.kind, wl.n, wl.noise, wl.policy, spl, "valu_issue_ratio"),
    }
    if wl.collector is not None:
        # the policy network's share: 10 240 flop per env and forward (5 layers of 32 x 32 x 2), one forward per step
        # plus the critic's two layers (4 096 flop) for the bootstrap value per launch; in the f16x3 form each product
        # costs three f16 MFMAs
        f32_equiv = wl.n * (10240.0 * spl + 4096.0) / (per_launch_ms * 1e-3) / 1e12
        split = wl.collector.policy.precision == "f16x3"
        peak = F16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
        r["mfma"] = {"form": wl.collector.policy.precision, "f32_equivalent_tflops": f32_equiv,
                     "achieved": f32_equiv * (3.0 if split else 1.0), "peak": peak, "unit": "TFLOP/s",
                     "frac": f32_equiv * (3.0 if split else 1.0) / peak}
    return r


def sub_run(M, torch, dev, spec, target_s=0.12):
    """One entry of configs_measured: a short timed region (~target_s, after the headline region has brought the clocks
    up) of another BASELINE.json configuration."""
    wl = Workload(M, torch, dev, **spec)
    try:
        K = wl.chunk
        wl.run(K)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        wl.run(4 * K)
        torch.cuda.synchronize(dev)
        per_launch = max(1e-6, (time.perf_counter() - t1) / 4)
        reps = max(8, int(target_s / per_launch))
        elapsed, event_ms, single = timed_region(torch, dev, wl, K, reps, singles=3)
        r = roofline_of(wl, K, reps, event_ms)
        assert wl.env.error_count() == 0
        out = {"workload": wl.describe(), "value": wl.n * K * reps / elapsed, "unit": "env-steps/s",
               "steps_per_launch": K, "timed_steps": K * reps, "timed_region_s": elapsed,
               "launch_ms": r["launch_ms"], "single_pass_ms": None if single is None else single * 1e3, "kernel": r["kernel"],
               "roofline": {k: r[k] for k in ("frac", "achieved", "fused_bytes_per_env_step", "frac_contract", "traffic")}}
        if "mfma" in r:
            out["roofline"]["mfma_frac"] = r["mfma"]["frac"]
            out["policy_products"] = r["mfma"]["form"]
        return out
    finally:
        wl.close()
        torch.cuda.empty_cache()


def exchange_leg(M, torch, dist, dev, args, world, rank, mode, force, barrier):
    """The rollout hand-off (BASELINE.json configs[4]): every rank rolls `--chunk` steps into one of two
    local buffer sets and hands the set to the learner over RCCL on a side stream while the next rollout steps into the
    other set (ShardedSortingEnv.rollout_async).  Same workload, same number of steps as the headline region."""
    from marl_sortingenv_amd.sharding import ShardedSortingEnv

    n, K = args.envs, max(1, min(args.chunk, args.steps))
    sh = ShardedSortingEnv(n * world, kind=args.kind, base_seed=0, device=dev, force_collective=force,
                           exchange_mode=mode, max_steps=args.max_steps, noise_sorting=args.noise, balesize=200,
                           auto_reset=True, rollout_pipeline=args.pipeline)
    tk = None
    for _ in range(3):                                    # warm-up: both buffer sets, both staging sets, RCCL channels
        tk = sh.rollout_async(K, policy_seed=2024)
        sh.collect(tk)
    torch.cuda.synchronize(dev)
    # size the region to ~0.5 s (over xGMI a hand-off of 7 x 150 MB takes milliseconds where a rollout takes 65 us)
    barrier()
    t1 = time.perf_counter()
    for _ in range(2):
        sh.collect(sh.rollout_async(K, policy_seed=2024))
    torch.cuda.synchronize(dev)
    cal = torch.tensor([(time.perf_counter() - t1) / 2], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(cal, op=dist.ReduceOp.MAX)    # every rank must run the same number of rollouts
    rollouts = int(max(8, min(-(-65536 // K), 0.5 / max(float(cal.item()), 1e-6))))
    # the gather alone, not overlapped: side-stream events around one hand-off of a finished rollout
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf = sh.env.rollout(K, policy_seed=2024, buffers=sh._sets[0])
    torch.cuda.synchronize(dev)
    barrier()
    side = sh.exchange.stream
    g0.record(side)
    t_alone = sh.exchange.gather_rollout_async(buf)
    g1.record(side)
    sh.exchange.wait(t_alone)
    torch.cuda.synchronize(dev)
    alone_ms = g0.elapsed_time(g1)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    prev = None
    for _ in range(rollouts):
        tk = sh.rollout_async(K, policy_seed=2024)
        if prev is not None:
            sh.collect(prev)                              # the learner's stream takes rollout i - 1 while i steps
        prev = tk
    sh.collect(prev)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, alone_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, alone_ms = float(t[0]), float(t[1])
    per_rank = sum(v.numel() * v.element_size() for v in buf.values() if v is not None)
    # bytes that cross xGMI into the busiest GPU per rollout: the learner takes every other rank's buffers (both modes);
    # with all-gather every rank does
    into_busiest = per_rank * (world - 1)
    sh.env.close()
    return {
        "mode": mode, "collective": "dist.gather -> rank 0 (rank-major staging)" if mode == "gather" else "dist.all_gather_into_tensor (rank-major)",
        "backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""),
        "world": world, "forced_at_world_1": bool(force and world == 1),
        "steps_per_rollout": K, "rollouts_timed": rollouts, "bytes_per_rollout_per_rank": per_rank,
        "bytes_into_learner_per_rollout": into_busiest,
        "ms": alone_ms, "GB/s_into_learner": (into_busiest / (alone_ms * 1e-3) / 1e9) if alone_ms > 0 and world > 1 else None,
        "ms_per_rollout_with_exchange": elapsed * 1e3 / rollouts,
        "value_with_exchange": n * world * K * rollouts / elapsed, "unit": "env-steps/s",
        "note": "side-stream collective overlapped with the next rollout's stepping (two local buffer sets, two staging sets); "
                "`ms` = one hand-off alone, not overlapped",
    }


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
    ap.add_argument("--cpu-budget", type=float, default=8.0)
    ap.add_argument("--no-configs", action="store_true", help="skip the configs_measured sub-runs (one GPU only)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "none", "allgather", "gather"],
                    help="rollout hand-off leg after the headline region: auto = gather to rank 0 when N > 1, none at N = 1")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: run the --exchange leg at N = 1 through a one-rank RCCL group")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import marl_sortingenv_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    # one rank per GPU over RCCL; with fewer GPUs than ranks (a rehearsal of the N > 1 path on a one-GPU box) the
    # ranks share devices and the collectives of this script go over gloo
    dev_index = (local_rank % max(1, n_dev)) if world > 1 else 0
    rehearsal = world > 1 and n_dev < world
    torch.cuda.set_device(dev_index)
    exchange_mode = args.exchange if args.exchange != "auto" else ("gather" if world > 1 else "none")
    forced = args.force_collective and world == 1 and exchange_mode != "none"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime

        # every collective of this script is short: a stuck one should end the run in minutes, not at the default
        kw = dict(timeout=datetime.timedelta(minutes=4))
        if forced:
            kw.update(rank=rank, world_size=world)
        if rehearsal:
            dist.init_process_group(backend="gloo", **kw)
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), **kw)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", dev_index)

    n = args.envs
    wl = Workload(M, torch, dev, kind=args.kind, n=n, chunk=args.chunk, noise=args.noise, max_steps=args.max_steps,
                  policy=args.policy, precision=args.precision, pipeline=args.pipeline, sort_policy=args.sort_policy,
                  mode=args.mode, no_outputs=args.no_outputs, index_offset=rank * n, rank=rank)

    def barrier():
        if world > 1:
            dist.barrier()

    # default: ~65 536 steps (0.2 s) in the timed region.  Shorter regions measure the clock ramp, not the kernel: at
    # --steps 20 a 8 ms region gives 75-77 us per launch, 70 ms 69.7, 0.27 s 68.6, 1.1 s 68.3
    reps = args.reps if args.reps > 0 else max(1, -(-65536 // args.steps))
    wl.run(args.warmup)
    torch.cuda.synchronize(dev)
    elapsed, event_ms, single = timed_region(torch, dev, wl, args.steps, reps, barrier)

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    roof = roofline_of(wl, args.steps, reps, event_ms) if args.mode != "step" else {
        "bound": "valu-issue", "priced_against": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
        "traffic": None, "kernel": "k_sample + k_step", "launch_ms": None}
    assert wl.env.error_count() == 0

    out = None
    if rank == 0:
        pol_tag = "random policy" if args.policy == "random" else f"learned policy, {wl.collector.policy.precision}"
        out = {
            "metric": f"env-steps/sec ({pol_tag}) Env_3_Monolith @65k envs" if args.kind == "mono" and n == 65536
                      else f"env-steps/sec ({pol_tag}) {KIND_NAME[args.kind]} @{n} envs",
            "value": n * world * args.steps * reps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "reps": reps,
            "timed_steps": args.steps * reps, "timed_region_s": elapsed,
            "ms_per_step": elapsed * 1e3 / (args.steps * reps),
            "single_pass_ms": None if single is None else single * 1e3,
            "single_pass_value": None if single is None else n * world * args.steps / single,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64 (integer state/masks, fp64 rewards, f32 observations)",
            "data": "synthetic",
            "config": {
                "workload": wl.describe().replace("masked-uniform random policy", "random-policy rollout"),
                "env_kind": args.kind, **({"sorting_policy": args.sort_policy} if args.kind == "press" else {}),
                "envs_per_gpu": n, "global_envs": n * world, "max_steps": args.max_steps,
                "balesize": 200, "noise_sorting": args.noise,
                "policy": "on-device masked-uniform, seed 2024" if args.policy == "random" else
                          f"actor-critic MLP 2x32 tanh (random-init weights, matrix products {wl.collector.policy.precision}) with masked categorical sampling, inside the rollout kernel",
                "steps_per_launch": min(wl.chunk, args.steps), "outputs": "none (diagnostic)" if args.no_outputs else
                "obs f32[K,N,D], mask u8[K,N,A], action i32, reward f32, done u8 per step",
                "parallelism": f"env-index sharding x{world}, no data-path collective in the timed region" +
                               (f" (REHEARSAL: {world} ranks on {n_dev} GPU)" if rehearsal else ""),
            },
            "roofline": roof,
        }

    # ---- N GPUs (or a forced one-rank rehearsal): the hand-off leg, after and apart from the headline region ----------
    if exchange_mode != "none" and (world > 1 or forced) and args.mode == "rollout" and args.policy == "random":
        wl.close()
        torch.cuda.empty_cache()
        try:
            ex = exchange_leg(M, torch, dist, dev, args, world, rank, exchange_mode, forced, barrier)
        except Exception as e:  # the headline stands without it; say what went wrong
            ex = {"mode": exchange_mode, "error": f"{type(e).__name__}: {e}"[:400]}
        if out is not None:
            out["exchange"] = ex
            out["exchange"]["hardware"] = "measured on this run's GPUs" if world > 1 and not rehearsal else \
                "rehearsal only (one rank / shared GPU): no xGMI transfer took place"
    elif out is not None and world > 1:
        out["exchange"] = {"mode": "none"}

    # ---- one GPU: BASELINE.json's other single-GPU configurations on the same clock ----------------------------------
    if rank == 0 and world == 1 and not args.no_configs and args.mode == "rollout":
        wl.close()
        torch.cuda.empty_cache()
        K = min(wl.chunk, args.steps)
        specs = [
            dict(kind="sort", n=65536, chunk=K),                                                   # configs[1]
            dict(kind="press", n=65536, chunk=K, sort_policy="frozen"),                            # configs[2]
            dict(kind="mono", n=262144, chunk=16),                                                 # configs[3], random
            dict(kind="mono", n=262144, chunk=16, policy="mlp", precision="f16x3"),                # configs[3], learned
            dict(kind="mono", n=262144, chunk=16, policy="mlp", precision="f32"),
            dict(kind="mono", n=65536, chunk=16, policy="mlp", precision="f16x3"),
            dict(kind="mono", n=65536, chunk=K, noise=0.05),                                       # the env classes' default noise
            dict(kind="mono", n=65536, chunk=64),                                                  # a 64-step collector
            dict(kind="mono", n=131072, chunk=64),
        ]
        measured = []
        for spec in specs:
            try:
                measured.append(sub_run(M, torch, dev, spec))
            except Exception as e:
                measured.append({"workload": json.dumps(spec), "error": f"{type(e).__name__}: {e}"[:300]})
        out["configs_measured"] = measured

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kind, args.max_steps, args.noise, 200, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
