"""The reference's policy-quality scenarios that need no trained model, at scale.

utils/benchmark_models.py:126-147 runs `Env_3_Monolith` for STEPS_TEST = 200 steps (main.py:42,50: noise 0) per seed and
sums the reward; utils/benchmark_plot_summary.py:6-17 quotes, over 10 seeds, Random -84.28 +- 22.29 and Rule-Based
44.03 +- 1.10 with masking, Random -109.36 +- 6.29 and Rule-Based 43.20 +- 1.07 without.  Here every seed 1..N is an env
lane: one fused rollout per scenario (rule-based and masked-uniform policies run on the device; the unmasked random
scenario steps with uniform actions over all 22).  Not a parity target (the reference's random mode draws from the
global np.random); the rule-based run is, and tests/test_gpu_api.py checks it seed by seed against the reference.

    python tools/benchmark_scenarios.py [--envs 65536]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import marl_sortingenv_amd as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
args = ap.parse_args()
n, T = args.envs, 200


def make():
    return M.BatchedSortingEnv(kind="mono", num_envs=n, device=0, base_seed=1, max_steps=T, noise_sorting=0.0,
                               balesize=200, auto_reset=True)  # exactly one episode is stepped


def total_reward(env, policy, masked=True):
    tot = torch.zeros((n,), dtype=torch.float64, device="cuda")
    if policy == "unmasked_random":
        g = torch.Generator(device="cuda").manual_seed(2024)
        for _ in range(T):
            a = torch.randint(0, env.num_actions, (n,), generator=g, device="cuda", dtype=torch.int32)
            _, rew, _, _ = env.step(a, use_action_masking=False, want_reward64=True)
            tot += env.reward64
        return tot
    K = 50
    buf = env.alloc_rollout(K, obs=False, mask=False)
    for _ in range(T // K):
        env.rollout(K, policy_seed=2024, buffers=buf, policy=policy, use_action_masking=masked)
        tot += buf["reward"].double().sum(dim=0)
    return tot


for label, policy, masked, quoted in (
        ("Random, masking", "random", True, "-84.28 +- 22.29"),
        ("Rule-Based, masking", "rule_based", True, "44.03 +- 1.10"),
        ("Random, no masking", "unmasked_random", False, "-109.36 +- 6.29")):
    r = total_reward(make(), policy, masked)
    print(f"{label:24s} {n} seeds: cumulative reward {float(r.mean()):8.2f} +- {float(r.std()):6.2f}   "
          f"(reference, 10 seeds: {quoted})")
print("Rule-Based, no masking   = the masked run: mode='rule_based' executes its action without validation whatever "
      "use_action_masking is (env_monolith.py:166-184, 262-264); the reference quotes 43.20 +- 1.07 for its second sample")
