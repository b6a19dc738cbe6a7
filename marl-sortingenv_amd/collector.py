"""On-device rollout collection with a learned policy (the caller side of the path).

SB3's `collect_rollouts` (the loop inside `model.learn`, src/training.py:191) alternates policy forward and
`env.step` on the host, one env at a time.  `PolicyRolloutCollector` does the same alternation for N envs with both
halves on the device - `mse_policy_forward` then `mse_step` - and fills buffers shaped like SB3's
`MaskableRolloutBuffer` ([K, N, ...]: observations, action masks, actions, log-probabilities, values, rewards,
episode starts), so a learner can consume them without a host copy.  For `Env_2_Pressing` an optional sorting policy
(13 -> 2, evaluated deterministically on `mse_sort_agent_obs`, env_2_press.py:101-104) plays the pre-trained sorting
agent.  Two launches per step.

`FusedPolicyRollout` is the same collection in ONE launch per rollout (`mse_rollout_policy`): the policy forward runs
inside the rollout kernel, the observation never leaves the wave's registers between the env transition and the
policy's MFMA chain.  Its buffers are bit-identical to `PolicyRolloutCollector`'s for the same seed.
"""
from __future__ import annotations

from typing import Optional

import torch

from .batched import BatchedSortingEnv
from .policy import MlpPolicy


class PolicyRolloutCollector:
    def __init__(self, env: BatchedSortingEnv, policy: MlpPolicy, n_steps: int, sort_policy: Optional[MlpPolicy] = None,
                 seed: int = 2024):
        if policy.obs_dim != env.obs_dim or policy.n_actions != env.num_actions:
            raise ValueError("policy dimensions do not match the env")
        if sort_policy is not None and (env.kind != "press" or sort_policy.obs_dim != 13 or sort_policy.n_actions != 2):
            raise ValueError("a sorting policy (13 -> 2) only applies to Env_2_Pressing")
        if not env.auto_reset:
            raise ValueError("collection needs auto_reset=True (episodes end inside a rollout)")
        self.env, self.policy, self.sort_policy = env, policy, sort_policy
        self.n_steps, self.seed, self.t = int(n_steps), int(seed), 0
        n, K, dev = env.num_envs, self.n_steps, env.device
        self.buffers = {
            "observations": torch.empty((K, n, env.obs_dim), dtype=torch.float32, device=dev),
            "action_masks": torch.empty((K, n, env.num_actions), dtype=torch.uint8, device=dev),
            "actions": torch.empty((K, n), dtype=torch.int32, device=dev),
            "log_probs": torch.empty((K, n), dtype=torch.float32, device=dev),
            "values": torch.empty((K, n), dtype=torch.float32, device=dev),
            "rewards": torch.empty((K, n), dtype=torch.float32, device=dev),
            "episode_starts": torch.empty((K, n), dtype=torch.uint8, device=dev),
        }
        self._last_done = torch.ones((n,), dtype=torch.uint8, device=dev)  # the envs were just reset
        self._out = None
        self._sort_out = None

    def collect(self, deterministic: bool = False) -> dict:
        """K steps; returns the buffers plus `last_values` f32[N] (the bootstrap value of the state after the last
        step) and `last_dones`, as SB3's compute_returns_and_advantage wants them."""
        env, b = self.env, self.buffers
        for k in range(self.n_steps):
            obs, mask = env.obs, env.mask
            b["observations"][k].copy_(obs)
            b["action_masks"][k].copy_(mask)
            b["episode_starts"][k].copy_(self._last_done)
            self._out = self.policy.forward(obs, mask, seed=self.seed, t=self.t, deterministic=deterministic,
                                            index_offset=env.index_offset, out=self._out)
            sort_mode = None
            if self.sort_policy is not None:
                self._sort_out = self.sort_policy.forward(env.sort_agent_obs(), None, deterministic=True, out=self._sort_out)
                sort_mode = self._sort_out["action"]
            b["actions"][k].copy_(self._out["action"])
            b["log_probs"][k].copy_(self._out["logp"])
            b["values"][k].copy_(self._out["value"])
            _, rew, done, _ = env.step(self._out["action"], sort_mode=sort_mode)
            b["rewards"][k].copy_(rew)
            self._last_done.copy_(done)
            self.t += 1
        last = self.policy.forward(env.obs, env.mask, seed=self.seed, t=self.t, deterministic=True,
                                   index_offset=env.index_offset)
        return dict(b, last_values=last["value"], last_dones=self._last_done.clone())


class FusedPolicyRollout:
    """K steps of policy forward + env transition per launch (`mse_rollout_policy`): MaskableRolloutBuffer-shaped
    device tensors [K, N, ...].  Env_2's sorting decisions: `sort_policy` (a 13 -> 2 MlpPolicy standing for the
    pre-trained sorting agent, evaluated inside the kernel), else `sort_mode` (a per-env tensor), else the reference's rule."""

    def __init__(self, env: BatchedSortingEnv, policy: MlpPolicy, n_steps: int, seed: int = 2024,
                 sort_mode: Optional[torch.Tensor] = None, sort_policy: Optional[MlpPolicy] = None):
        if policy.obs_dim != env.obs_dim or policy.n_actions != env.num_actions:
            raise ValueError("policy dimensions do not match the env")
        if sort_policy is not None and (env.kind != "press" or sort_policy.obs_dim != 13 or sort_policy.n_actions != 2):
            raise ValueError("a sorting policy (13 -> 2) only applies to Env_2_Pressing")
        if not env.auto_reset:
            raise ValueError("collection needs auto_reset=True (episodes end inside a rollout)")
        self.env, self.policy, self.n_steps, self.seed = env, policy, int(n_steps), int(seed)
        self.sort_policy = sort_policy  # Env_2's pre-trained sorting agent, evaluated inside the kernel
        self.sort_mode = None if sort_mode is None else sort_mode.to(device=env.device, dtype=torch.int32).contiguous()
        n, K, dev = env.num_envs, self.n_steps, env.device
        self.buffers = {
            "observations": torch.empty((K, n, env.obs_dim), dtype=torch.float32, device=dev),
            "action_masks": torch.empty((K, n, env.num_actions), dtype=torch.uint8, device=dev),
            "actions": torch.empty((K, n), dtype=torch.int32, device=dev),
            "log_probs": torch.empty((K, n), dtype=torch.float32, device=dev),
            "values": torch.empty((K, n), dtype=torch.float32, device=dev),
            "rewards": torch.empty((K, n), dtype=torch.float32, device=dev),
            "episode_starts": torch.empty((K, n), dtype=torch.uint8, device=dev),
            "last_values": torch.empty((n,), dtype=torch.float32, device=dev),
            "last_dones": torch.empty((n,), dtype=torch.uint8, device=dev),
        }

    def collect(self, n_steps: Optional[int] = None, deterministic: bool = False, use_action_masking: bool = True,
                check_overflow: bool = False) -> dict:
        """One launch.  n_steps <= the buffers' K (a shorter rollout fills the first rows).  The env's own obs / mask
        tensors are not refreshed (the state is; call env.refresh_outputs() to step it by hand afterwards)."""
        import ctypes as C

        from ._lib import MSE_STEP_CHECK_OVERFLOW, MSE_STEP_UNMASKED, check

        K = self.n_steps if n_steps is None else int(n_steps)
        if not 1 <= K <= self.n_steps:
            raise ValueError("n_steps must be in [1, the collector's K]")
        env, b = self.env, self.buffers
        flags = (0 if use_action_masking else MSE_STEP_UNMASKED) | (MSE_STEP_CHECK_OVERFLOW if check_overflow else 0)

        def ptr(x):
            return None if x is None else C.c_void_p(x.data_ptr())

        with torch.cuda.device(env.device):
            check(env.L.mse_rollout_policy(env._h, self.policy._h, None if self.sort_policy is None else self.sort_policy._h,
                                           K, self.seed, 1 if deterministic else 0,
                                           ptr(self.sort_mode), flags, ptr(b["observations"]), ptr(b["action_masks"]),
                                           ptr(b["actions"]), ptr(b["log_probs"]), ptr(b["values"]), ptr(b["rewards"]),
                                           ptr(b["episode_starts"]), ptr(b["last_values"]), ptr(b["last_dones"]),
                                           env._stream()))
        return b
