"""Single-env Gymnasium views: the reference's three env classes as N=1 handles on the HIP engine.

Same constructor, `reset(seed) -> (obs, {})`, `step(action, ...) -> (obs, float, bool, False, info)`,
`action_masks()`, `observation_space`, `action_space`, `.name`, `set_agents()` as the reference
(src/envs_train/env_1_sort.py, env_2_press.py, env_monolith.py), so ActionMasker / Monitor /
check_env / PPO / MaskablePPO accept them unchanged.  NumPy in/out on the host (one tiny
device->host copy per call): this view is for drop-in compatibility, the batched front-end
(batched.py) is the fast path.  gymnasium is optional and imported lazily.
"""
from __future__ import annotations

import numpy as np
import torch

from .batched import BatchedSortingEnv
from .config import NUM_ACTIONS, OBS_DIM, SortingEnvConfig

try:  # pragma: no cover - not installed in the build container
    import gymnasium as _gym
    from gymnasium import spaces as _spaces

    _EnvBase = _gym.Env
except Exception:  # duck-typed stand-ins with the attributes SB3 reads
    _gym = None

    class _EnvBase:
        metadata = {"render_modes": []}
        render_mode = None
        spec = None

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class _Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def seed(self, seed=None):
            return [seed]

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class _Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def seed(self, seed=None):
            return [seed]

        def contains(self, x):
            return 0 <= int(x) < self.n

    class _spaces:  # noqa: N801
        Box = _Box
        Discrete = _Discrete


def _sort_bounds():
    low = np.concatenate([np.zeros(9), np.full(4, -1.0)])
    return low, np.ones(13)


class _SingleEnv(_EnvBase):
    kind = "mono"

    def __init__(self, max_steps: int = 50, seed: int = None, noise_sorting: float = 0.05, balesize: int = 200,
                 simulation=False, config: SortingEnvConfig | None = None, device=0):
        self.name = self.kind
        self.max_steps = max_steps
        self.seed = seed or 0
        self.noise_accuracy = noise_sorting
        self.bale_standard_size = balesize
        self._batched = BatchedSortingEnv(kind=self.kind, num_envs=1, device=device, base_seed=self.seed,
                                          max_steps=max_steps, noise_sorting=noise_sorting, balesize=balesize,
                                          config=config, auto_reset=False)
        if seed is None:
            # the reference seeds the streams with 0 but leaves the generator unseeded (env_super.py:110,167)
            self._batched.reset(seeds=None)
        self._initialize_spaces()
        self._action = torch.zeros(1, dtype=torch.int32, device=self._batched.device)
        self._sort_mode = torch.zeros(1, dtype=torch.int32, device=self._batched.device)
        self.current_step = 0

    # -- spaces (env_1_sort.py:43-72, env_2_press.py:44-64, env_monolith.py:49-79) -------------
    def _initialize_spaces(self):
        if self.kind == "sort":
            low, high = _sort_bounds()
        elif self.kind == "press":
            low, high = np.zeros(16), np.ones(16)
        else:
            sl, sh = _sort_bounds()
            low, high = np.concatenate([sl, np.zeros(16)]), np.concatenate([sh, np.ones(16)])
        self.observation_space = _spaces.Box(low.astype(np.float32), high.astype(np.float32), dtype=np.float32)
        self.action_space = _spaces.Discrete(NUM_ACTIONS[self.kind])

    def action_masks(self):
        return self._batched.action_masks()[0].cpu().numpy().astype(bool)

    def get_obs(self):
        return self._batched.obs[0].cpu().numpy()

    def reset(self, *, seed=None, options=None):
        seeds = None if seed is None else torch.tensor([int(seed)], dtype=torch.int64)
        obs, _ = self._batched.reset(seeds=seeds)
        if seed is not None:
            self.seed = seed or 0
        self.current_step = 0
        return obs[0].cpu().numpy(), {}

    def _step(self, action, sort_mode, use_action_masking, check_overflow):
        if action is None:
            raise ValueError("Invalid action source: provide 'action' (the in-env agent/rule modes of the "
                             "reference's eval loop are not part of the step path)")
        self._action[0] = int(action)
        sm = None
        if sort_mode is not None:
            self._sort_mode[0] = int(sort_mode)
            sm = self._sort_mode
        obs, rew, done, _ = self._batched.step(self._action, sort_mode=sm, use_action_masking=use_action_masking,
                                               check_overflow=check_overflow, want_reward64=True)
        self.current_step += 1
        reward = float(self._batched.reward64[0].item())
        terminated = bool(done[0].item())
        return obs[0].cpu().numpy(), reward, terminated, False

    def get_state(self):
        ints, dbls, rng = self._batched.get_state()
        return ints[0].cpu().numpy(), dbls[0].cpu().numpy(), rng[0].cpu().numpy().view(np.uint64)

    def close(self):
        self._batched.close()


class Env_1_Sorting(_SingleEnv):
    """Sorting agent only: Discrete(2) sensor mode, 13-dim obs (reference env_1_sort.py)."""
    kind = "sort"

    def set_agents(self, press_agent=None):
        self.press_agent = press_agent

    def reset(self, seed=None):  # the reference's Env_1.reset has no `options` (env_1_sort.py:81)
        return super().reset(seed=seed)

    def step(self, action=None, use_action_masking=True, check_overflow=False):
        obs, reward, terminated, truncated = self._step(action, None, use_action_masking, check_overflow)
        return obs, reward, terminated, truncated, {"action": action}


class Env_2_Pressing(_SingleEnv):
    """Pressing agent only: Discrete(11), 16-dim obs (reference env_2_press.py).  The sorting
    decision comes from `sort_agent.predict` if set (env_2_press.py:101-104), else from the
    rule-based fallback evaluated on the device."""
    kind = "press"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.sort_agent = None

    def set_agents(self, sort_agent=None):
        self.sort_agent = sort_agent

    def step(self, action, use_action_masking=True, check_overflow=False):
        sort_mode = None
        if self.sort_agent is not None:
            # the agent sees get_sort_obs() AFTER this step's material flow (env_2_press.py:101-104):
            # mse_sort_agent_obs previews exactly that view of the coming step
            sort_obs = self._batched.sort_agent_obs()[0].cpu().numpy()
            sort_mode, _ = self.sort_agent.predict(sort_obs, deterministic=True)
            sort_mode = int(sort_mode)
        obs, reward, terminated, truncated = self._step(action, sort_mode, use_action_masking, check_overflow)
        return obs, reward, terminated, truncated, {"action": int(action)}


class Env_3_Monolith(_SingleEnv):
    """Joint sorting + pressing: Discrete(22) = mode*11 + press action, 29-dim obs
    (reference env_monolith.py)."""
    kind = "mono"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.sort_agent = self.press_agent = self.mono_agent = None

    def set_agents(self, sort_agent=None, press_agent=None, mono_agent=None):
        self.sort_agent, self.press_agent, self.mono_agent = sort_agent, press_agent, mono_agent

    def step(self, action=None, mode=None, use_action_masking=True, check_overflow=False):
        if action is None and self.mono_agent is not None:  # env_monolith.py:144-150
            action, _ = self.mono_agent.predict(self.get_obs(), deterministic=True, action_masks=self.action_masks())
        obs, reward, terminated, truncated = self._step(action, None, use_action_masking, check_overflow)
        return obs, reward, terminated, truncated, {"action": int(action)}
