"""SB3 VecEnv-protocol adapter over the batched HIP engine (duck-typed: stable-baselines3 is optional).

Implements what SB3's on-policy loop calls (stable_baselines3 VecEnv): num_envs, observation_space,
action_space, reset(), step_async()/step_wait()/step(), env_method("action_masks"), get_attr,
set_attr, env_is_wrapped, seed(), close(), and infos[i]["terminal_observation"] on auto-reset.
SB3's contract is NumPy on the host, so each step does one device->host copy of obs/reward/done;
the zero-copy path is BatchedSortingEnv / its rollout().  This is the consumer side of the
reference's boundary (src/training.py:58-71,118-143), not a reimplementation of SB3.
"""
from __future__ import annotations

import numpy as np
import torch

from .batched import BatchedSortingEnv
from .envs import _spaces, _sort_bounds
from .config import NUM_ACTIONS

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
except Exception:
    _VecEnvBase = object


class SortingVecEnv(_VecEnvBase):
    def __init__(self, kind="mono", num_envs=1024, device=0, base_seed=0, max_steps=50, noise_sorting=0.05,
                 balesize=200, config=None, use_action_masking=True):
        self.env = BatchedSortingEnv(kind=kind, num_envs=num_envs, device=device, base_seed=base_seed,
                                     max_steps=max_steps, noise_sorting=noise_sorting, balesize=balesize,
                                     config=config, auto_reset=True)
        self.kind = kind
        self.use_action_masking = use_action_masking
        if kind == "sort":
            low, high = _sort_bounds()
        elif kind == "press":
            low, high = np.zeros(16), np.ones(16)
        else:
            sl, sh = _sort_bounds()
            low, high = np.concatenate([sl, np.zeros(16)]), np.concatenate([sh, np.ones(16)])
        observation_space = _spaces.Box(low.astype(np.float32), high.astype(np.float32), dtype=np.float32)
        action_space = _spaces.Discrete(NUM_ACTIONS[kind])
        if _VecEnvBase is not object:
            super().__init__(num_envs, observation_space, action_space)
        else:
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
        self._actions = None
        self.render_mode = None

    def reset(self):
        obs, _ = self.env.reset(seeds=self.env.seeds)
        return obs.cpu().numpy()

    def seed(self, seed=None):
        if seed is not None:
            self.env.seeds = torch.arange(self.num_envs, dtype=torch.int64, device=self.env.device) + int(seed)
        return [int(s) for s in self.env.seeds[: min(self.num_envs, 8)].tolist()]

    def step_async(self, actions):
        self._actions = torch.as_tensor(np.asarray(actions), dtype=torch.int32)

    def step_wait(self):
        obs, rew, done, _ = self.env.step(self._actions, use_action_masking=self.use_action_masking,
                                          want_terminal_obs=True)
        obs_h, rew_h = obs.cpu().numpy(), rew.cpu().numpy()
        done_h = done.cpu().numpy().astype(bool)
        infos = [{} for _ in range(self.num_envs)]
        idx = np.flatnonzero(done_h)
        if idx.size:
            term = self.env.terminal_obs[torch.as_tensor(idx, device=self.env.device)].cpu().numpy()
            for k, i in enumerate(idx):
                infos[i]["terminal_observation"] = term[k]
                infos[i]["TimeLimit.truncated"] = False
        return obs_h, rew_h, done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def action_masks(self):
        return self.env.action_masks().cpu().numpy().astype(bool)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        if method_name == "action_masks":
            m = self.action_masks()
            idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
            return [m[i] for i in idx]
        raise AttributeError(f"env_method({method_name!r}) is not provided by the batched engine")

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else (1 if isinstance(indices, int) else len(indices))
        if attr_name == "name":
            return [self.kind] * n
        if attr_name == "render_mode":
            return [None] * n
        raise AttributeError(attr_name)

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError(f"set_attr({attr_name!r}) is not supported")

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else (1 if isinstance(indices, int) else len(indices))
        return [False] * n

    def close(self):
        self.env.close()
