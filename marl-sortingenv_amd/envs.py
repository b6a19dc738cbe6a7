"""Single-env Gymnasium views: the reference's three env classes as N=1 handles on the HIP engine.

Same constructor, `reset(seed) -> (obs, {})`, `step(action, ...) -> (obs, float, bool, False, info)`,
`action_masks()`, `observation_space`, `action_space`, `.name`, `set_agents()` as the reference
(src/envs_train/env_1_sort.py, env_2_press.py, env_monolith.py), so ActionMasker / Monitor /
check_env / PPO / MaskablePPO accept them unchanged, `copy.deepcopy(env.unwrapped)` gives an independent
eval env (src/training.py:82) and `test_env`'s `step(action=None, mode=...)` loop runs (src/testing.py:53).
NumPy in/out on the host (a few tiny device->host copies per call): this view is for drop-in compatibility,
the batched front-end (batched.py) is the fast path.  gymnasium is optional and imported lazily.

Every transition is computed by the device engine.  What lives here is the host-side glue the reference keeps in
Python too: which source the action comes from (agent objects, the global `np.random` of mode='random'), and the
dashboard ledgers rebuilt from the engine's one-env trace (trace.py).
"""
from __future__ import annotations

import random as _py_random

import numpy as np
import torch

from .batched import BatchedSortingEnv
from .config import NUM_ACTIONS, OBS_DIM, SortingEnvConfig
from .trace import ACC_BELT, INTERNAL, MATERIALS, OVERFLOW, EnvTrace

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


_TRACE_CAPACITY = 1024


class _SingleEnv(_EnvBase):
    kind = "mono"

    def __init__(self, max_steps: int = 50, seed: int = None, noise_sorting: float = 0.05, balesize: int = 200,
                 simulation=False, config: SortingEnvConfig | None = None, device=0, _state=None):
        self.name = self.kind
        self.max_steps = max_steps
        self.seed = seed or 0
        self.noise_accuracy = noise_sorting
        self.bale_standard_size = balesize
        self._ctor = dict(max_steps=max_steps, seed=seed, noise_sorting=noise_sorting, balesize=balesize,
                          simulation=simulation, config=config, device=device)
        self._batched = BatchedSortingEnv(kind=self.kind, num_envs=1, device=device, base_seed=self.seed,
                                          max_steps=max_steps, noise_sorting=noise_sorting, balesize=balesize,
                                          config=config, auto_reset=False, reset_now=_state is None)
        cfg = self._batched.config
        self.press_times = {1: cfg.press_times[0], 2: cfg.press_times[1]}
        self.quality_thresholds = dict(zip("ABCD", cfg.bale_quality_thresholds))
        self.container_global_max = cfg.container_capacity
        self.material_names = ["A", "B", "C", "D"]
        self._baseline_accuracy = [float(a) for a in cfg.baseline_accuracy]
        if _state is not None:
            self._batched.set_state(*_state)          # a copy of another env (__deepcopy__)
            self._batched.refresh_outputs()
        elif seed is None:
            # the reference seeds the streams with `seed or 0` but builds its generator from OS entropy
            # (env_super.py:110,167): streams as reset(0), pattern order by the build's unseeded rule
            self._batched.reset(seeds=None)
        self._initialize_spaces()
        self._action = torch.zeros(1, dtype=torch.int32, device=self._batched.device)
        self._sort_mode = torch.zeros(1, dtype=torch.int32, device=self._batched.device)
        self.current_step = 0
        self._overflow = None
        self._trace = EnvTrace(balesize if balesize is not None else cfg.bale_standard_size, cfg.bale_remainder_threshold)
        self._batched.trace_begin(0, _TRACE_CAPACITY)

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
            # env_super.py:165-184 set_seed: besides the env's own streams (seeded on the device) the reference seeds
            # the process-global generators, which its mode='random' draws from (env_monolith.py:158,162)
            self.seed = seed or 0
            np.random.seed(self.seed)
            _py_random.seed(self.seed)
            self.action_space.seed(self.seed)
            self.observation_space.seed(self.seed)
        self.current_step = 0
        self._trace.reset()  # env_super.py:392,400-408: the ledgers restart with the episode
        self._batched.trace_begin(0, _TRACE_CAPACITY)
        return obs[0].cpu().numpy(), {}

    def _step(self, action, sort_mode, use_action_masking, check_overflow, sanitize_late=False):
        if action is None:
            raise ValueError("Invalid action source: Provide 'action', set 'mode' to 'random', 'rule_based', or "
                             "'model', or assign a mono_agent.")
        if self._batched._trace_n >= _TRACE_CAPACITY:  # an env stepped on past its episode: drain and go on
            self._drain_trace()
        self._action[0] = int(action)
        sm = None
        if sort_mode is not None:
            self._sort_mode[0] = int(sort_mode)
            sm = self._sort_mode
        obs, rew, done, _ = self._batched.step(self._action, sort_mode=sm, use_action_masking=use_action_masking,
                                               check_overflow=check_overflow, want_reward64=True,
                                               sanitize_late=sanitize_late)
        self.current_step += 1
        reward = float(self._batched.reward64[0].item())
        terminated = bool(done[0].item())
        self._overflow = None
        if check_overflow and terminated:
            # env_monolith.py:264-268, env_1_sort.py:131-139, env_2_press.py:144-150: an overflow termination reports
            # the material detect_overflow() found (env_super.py:900-905); the engine's trace record carries it
            rec = self._batched.trace_records()[-1].cpu().numpy()
            if int(rec[OVERFLOW]) > 0:
                self._overflow = (MATERIALS[int(rec[OVERFLOW]) - 1], int(rec[INTERNAL]))
        return obs[0].cpu().numpy(), reward, terminated, False

    def _info(self, action):
        """The step's info dict: {"action": a}, with the reference's overflow keys in front when check_overflow
        ended the episode."""
        if self._overflow is not None:
            return {"overflow": True, "overflow_material": self._overflow[0], "action": action}
        return {"action": action}

    # -- copy.deepcopy(env.unwrapped) (src/training.py:82): an independent env in the same state --------------------
    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        memo[id(self)] = new
        self._drain_trace()
        state = self._batched.get_state()
        _SingleEnv.__init__(new, **self._ctor, _state=state)
        if self.kind != "sort":
            new.sort_agent = getattr(self, "sort_agent", None)   # agents are shared, not copied: the caller
        if self.kind != "press":                                  # detaches / re-attaches them around the copy
            new.press_agent = getattr(self, "press_agent", None)
        if self.kind == "mono":
            new.mono_agent = getattr(self, "mono_agent", None)
        new.seed, new.current_step = self.seed, self.current_step
        new._batched.policy_step = self._batched.policy_step
        new._trace.extend(self._trace.records) if self._trace.records else None
        return new

    def __copy__(self):
        return self.__deepcopy__({})

    # -- the reference's per-env ledgers (utils/plotting.py:32-48 reads them off the env) --------------------------
    def _drain_trace(self):
        rec = self._batched.trace_records()
        if rec.shape[0]:
            self._trace.extend(rec.cpu().numpy())
            self._batched.trace_begin(0, _TRACE_CAPACITY)
        return self._trace

    @property
    def reward_data(self):
        return self._drain_trace().reward_data

    @property
    def press_actions_per_timestep(self):
        return self._drain_trace().press_actions_per_timestep

    @property
    def bale_count(self):
        return self._drain_trace().bale_count

    def _snapshot(self):
        ints, dbls, _ = self._batched.get_state()
        return ints[0].cpu().numpy(), dbls[0].cpu().numpy()

    @property
    def current_material_input(self):
        return [int(v) for v in self._snapshot()[0][0:4]]

    @property
    def current_material_belt(self):
        return [int(v) for v in self._snapshot()[0][4:8]]

    @property
    def current_material_sorting(self):
        return [int(v) for v in self._snapshot()[0][8:12]]

    @property
    def container_materials(self):
        i = self._snapshot()[0]
        out = {m: int(i[12 + k]) for k, m in enumerate("ABCD")}
        out.update({f"{m}_False": int(i[16 + k]) for k, m in enumerate("ABCD")})
        out["E"] = int(i[20])
        return out

    @property
    def press_state(self):
        i = self._snapshot()[0]
        out = {}
        for p in (1, 2):
            mat = int(i[23 + p - 1])
            out[f"press_{p}"] = int(i[21 + p - 1])
            out[f"material_{p}"] = MATERIALS[mat] if mat >= 0 else 0
            out[f"n_{p}"] = int(i[25 + p - 1])
            out[f"q_{p}"] = int(i[27 + p - 1]) / 100.0
        return out

    @property
    def accuracy_belt(self):
        return [float(v) for v in self._snapshot()[1]]

    @property
    def accuracy_sorter(self):
        """accuracy_belt as it stood before the last step's update (env_super.py:457)."""
        recs = self._drain_trace().records
        if len(recs) >= 2:
            return [float(v) for v in recs[-2][ACC_BELT:ACC_BELT + 4]]
        return list(self._baseline_accuracy)

    @property
    def sensor_current_setting(self):
        return int(self._snapshot()[0][29])

    @property
    def belt_occupancy(self):
        belt = self.current_material_belt
        return round(sum(belt) / 100, 2) if sum(belt) else 0.0

    def render(self, *args, **kwargs):
        """The reference's dashboard (utils/plotting.py) is outside the step path; the attributes it reads are here,
        so `plot_env(env)` can be called by whoever has it."""
        return None

    def get_state(self):
        ints, dbls, rng = self._batched.get_state()
        return ints[0].cpu().numpy(), dbls[0].cpu().numpy(), rng[0].cpu().numpy().view(np.uint64)

    def close(self):
        self._batched.close()


class Env_1_Sorting(_SingleEnv):
    """Sorting agent only: Discrete(2) sensor mode, 13-dim obs (reference env_1_sort.py)."""
    kind = "sort"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.press_agent = None

    def set_agents(self, press_agent=None):
        self.press_agent = press_agent

    def reset(self, seed=None):  # the reference's Env_1.reset has no `options` (env_1_sort.py:81)
        return super().reset(seed=seed)

    def step(self, action=None, use_action_masking=True, check_overflow=False):
        obs, reward, terminated, truncated = self._step(action, None, use_action_masking, check_overflow)
        if self._overflow is not None:
            # env_1_sort.py:125,139: on overflow "action" is the press job the env sampled itself, in
            # press_discrete_to_action's form [press_id, material_id] / [0, None] (env_super.py:804-809)
            a = self._overflow[1]
            return obs, reward, terminated, truncated, self._info([0, None] if a == 0 else [1 if a <= 5 else 2, (a - 1) % 5])
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
        return obs, reward, terminated, truncated, self._info(int(action))


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
        """The reference's action-source hierarchy (env_monolith.py:109-224): an explicit action, else the stored
        monolith agent, else `mode`: 'random' | 'rule_based' | 'model'."""
        masked, late = use_action_masking, False
        if action is not None:
            action = int(action)
        elif self.mono_agent is not None:                                   # env_monolith.py:144-150
            # the agent sees get_obs() AFTER this step's flow update (:113-114, 145 -> :98-104): the two previews of
            # the coming step, concatenated; the mask does not depend on the flow update
            obs = np.concatenate([self._batched.sort_agent_obs()[0].cpu().numpy(),
                                  self._batched.press_agent_obs()[0].cpu().numpy()])
            action, _ = self.mono_agent.predict(obs, deterministic=True, action_masks=self.action_masks())
            action, masked = int(action), True                              # applied without sanitising (:254-257)
        elif mode == "random":                                              # env_monolith.py:152-164
            # the reference draws from the process-global np.random, which reset(seed) seeds (env_super.py:177)
            if use_action_masking:
                valid = np.flatnonzero(self.action_masks())
                action = int(np.random.choice(valid)) if valid.size > 0 else 0
            else:
                action = int(np.random.randint(0, self.action_space.n))
                late = True                                                 # sanitised after the sort (:245-253)
        elif mode == "rule_based":                                          # env_monolith.py:166-184
            action, masked = int(self._batched.rule_actions()[0].item()), True
        elif mode == "model":                                               # env_monolith.py:186-221
            action, masked = self._model_action(use_action_masking), True
        else:
            raise ValueError("Invalid action source: Provide 'action', set 'mode' to 'random', 'rule_based', or "
                             "'model', or assign a mono_agent.")
        obs, reward, terminated, truncated = self._step(action, None, masked, check_overflow, sanitize_late=late)
        return obs, reward, terminated, truncated, self._info(action)

    def _model_action(self, use_action_masking):
        """Modular agents, each falling back to the env's own random draw when absent (env_monolith.py:186-221)."""
        drawn = int(self._batched.model_actions(use_action_masking, draw_sort=self.sort_agent is None,
                                                draw_press=self.press_agent is None)[0].item())
        sort_mode, press_action = drawn // 11, drawn % 11
        if self.sort_agent is not None:
            sort_obs = self._batched.sort_agent_obs()[0].cpu().numpy()      # get_sort_obs() after the flow update
            sort_mode = int(self.sort_agent.predict(sort_obs, deterministic=True)[0])
        if self.press_agent is not None:
            press_obs = self._batched.press_agent_obs()[0].cpu().numpy()
            is_maskable = hasattr(self.press_agent, "policy") and "Maskable" in str(type(self.press_agent))
            if use_action_masking and is_maskable:
                mask = self.action_masks()[:11]                             # press_action_masks()
                press_action = int(self.press_agent.predict(press_obs, deterministic=True, action_masks=mask)[0])
            else:
                press_action = int(self.press_agent.predict(press_obs, deterministic=True)[0])
        return sort_mode * 11 + press_action
