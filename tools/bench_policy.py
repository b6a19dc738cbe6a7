"""Throughput of mse_policy_forward and of the policy -> step loop.

    python tools/bench_policy.py [--envs 65536] [--iters 200]

Prints one JSON line: rows/s of the policy forward alone (HIP events around the run of launches), its roofline
(f32 MFMA: 80 v_mfma_f32_32x32x2_f32 per tile of 32 rows = 10 240 flop per row, padding included; peak 157.3 TFLOP/s,
MI355X_MICROARCH.md), and env-steps/s of  forward -> mse_step  with the MLP policy acting."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import marl_sortingenv_amd as M  # noqa: E402
from marl_sortingenv_amd.policy import SB3_KEYS, _shapes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--iters", type=int, default=200)
args = ap.parse_args()
n = args.envs
env = M.BatchedSortingEnv(kind="mono", num_envs=n, device=0, base_seed=0, max_steps=200, noise_sorting=0.0)
g = torch.Generator().manual_seed(0)
w = {k: (torch.randn(s, generator=g) * 0.3).float() for k, s in zip(SB3_KEYS, _shapes(env.obs_dim, env.num_actions))}
pol = M.MlpPolicy(env.obs_dim, env.num_actions, w, device=0)
obs, mask = env.obs, env.mask
buf = None
for t in range(10):
    buf = pol.forward(obs, mask, t=t, out=buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(args.iters):
    pol.forward(obs, mask, t=t, out=buf)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
flop_per_row = 80 * 32 * 32 * 2 * 2 / 32
e0.record()
for t in range(args.iters):
    out = pol.forward(obs, mask, t=t, out=buf)
    obs, rew, done, mask = env.step(out["action"])
e1.record()
torch.cuda.synchronize()
ms_loop = e0.elapsed_time(e1) / args.iters
print(json.dumps({
    "metric": "policy rows/sec (MLP 29-32-32-22 + value head, masked sampling)", "value": n / (ms * 1e-3), "unit": "rows/s",
    "envs": n, "ms_per_forward": ms,
    "roofline": {"bound": "mfma", "achieved": flop_per_row * n / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                 "frac": flop_per_row * n / (ms * 1e-3) / 1e12 / 157.3, "flop_per_row": flop_per_row,
                 "hbm_bytes_per_row": 4 * env.obs_dim + env.num_actions + 12},
    "policy_step_loop": {"env_steps_per_s": n / (ms_loop * 1e-3), "ms_per_step": ms_loop,
                         "what": "mse_policy_forward + mse_step per step, all on device"},
}))
