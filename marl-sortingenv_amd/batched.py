"""Torch-native batched front-end of the HIP step engine: N env instances, one per GPU lane.

This is the zero-host-copy surface (`step(actions) -> (obs, reward, done, mask)` on device tensors)
that the single-env Gymnasium views (envs.py) and the SB3 VecEnv adapter (vec_env.py) sit on.
PyTorch is plumbing here: device memory for the I/O buffers and the HIP stream; every transition
is computed by libmse_hip.so through the C ABI of include/mse.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (MSE_MODEL_NO_PRESS_DRAW, MSE_MODEL_NO_SORT_DRAW, MSE_ROLLOUT_RULE_BASED, MSE_SNAP_INTS, MSE_SNAP_RNG_WORDS,
                   MSE_STEP_CHECK_OVERFLOW, MSE_STEP_SANITIZE_LATE, MSE_STEP_UNMASKED, MSE_TRACE_COLS, check, load_library)
from .config import NUM_ACTIONS, OBS_DIM, SortingEnvConfig


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedSortingEnv:
    """N independent Env_1_Sorting / Env_2_Pressing / Env_3_Monolith instances on one MI355X.

    kind: "sort" | "press" | "mono" (the reference's env.name values).
    Env i is seeded like the reference's `Env(seed=base_seed + index_offset + i)` followed by
    `reset(seed=...)`: results do not depend on how envs are sharded over GPUs.
    """

    def __init__(self, kind: str = "mono", num_envs: int = 1, device: int | str | torch.device = 0,
                 base_seed: int = 0, seeds: Optional[torch.Tensor] = None, max_steps: int = 50,
                 noise_sorting: Optional[float] = 0.05, balesize: Optional[int] = 200,
                 config: Optional[SortingEnvConfig] = None, auto_reset: bool = True,
                 track_bales: bool = True, literal_choice: bool = False, index_offset: int = 0,
                 reset_now: bool = True, rollout_pipeline: int = 0, library: Optional[str] = None):
        if kind not in OBS_DIM:
            raise ValueError(f"kind must be one of {sorted(OBS_DIM)}")
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedSortingEnv needs a HIP device: the step path has no CPU fallback")
        self.L = load_library(library)  # `library`: another build of the same source (tests)
        self.kind = self.name = kind
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.config = config if config is not None else SortingEnvConfig()
        self.max_steps = int(max_steps)
        self.index_offset = int(index_offset)
        self.obs_dim, self.num_actions = OBS_DIM[kind], NUM_ACTIONS[kind]
        self._cfg_struct = self.config.to_struct(kind, max_steps, noise_sorting, balesize, auto_reset,
                                                 track_bales, literal_choice, rollout_pipeline)
        self.auto_reset = bool(auto_reset)
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        check(self.L.mse_create_indexed(C.byref(h), C.byref(self._cfg_struct), self.num_envs, dev_index,
                                        self.index_offset))
        self._h = h
        self._trace_buf, self._trace_n = None, 0
        n, dev = self.num_envs, self.device
        self.obs = torch.zeros((n, self.obs_dim), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.reward64 = torch.zeros((n,), dtype=torch.float64, device=dev)
        self.done = torch.zeros((n,), dtype=torch.uint8, device=dev)
        self.mask = torch.zeros((n, self.num_actions), dtype=torch.uint8, device=dev)
        self.terminal_obs = torch.zeros((n, self.obs_dim), dtype=torch.float32, device=dev)
        if seeds is None:
            seeds = (torch.arange(n, dtype=torch.int64, device=dev) + (int(base_seed) + self.index_offset))
        self.seeds = seeds.to(device=dev, dtype=torch.int64).contiguous()
        if reset_now:
            self.reset(seeds=self.seeds)

    # ---- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.L.mse_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- reference surface, batched ----------------------------------------------------------
    def reset(self, seeds: Optional[torch.Tensor] = None, which: Optional[torch.Tensor] = None):
        """reset(seed=seeds[i]) for every env (or those with which[i] != 0); seeds=None is the
        reference's reset(seed=None): streams continue.  Returns (obs, mask) device tensors."""
        if seeds is not None:
            seeds = seeds.to(device=self.device, dtype=torch.int64).contiguous()
            if seeds.numel() != self.num_envs:
                raise ValueError("seeds must have one entry per env")
            if bool((seeds < 0).any()):
                raise ValueError("seeds must be non-negative")  # np.random.default_rng rejects negatives
        if which is not None:
            which = which.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            check(self.L.mse_reset(self._h, _ptr(seeds), _ptr(which), _ptr(self.obs), _ptr(self.mask), self._stream()))
        return self.obs, self.mask

    def step(self, actions: torch.Tensor, sort_mode: Optional[torch.Tensor] = None,
             use_action_masking: bool = True, check_overflow: bool = False, want_reward64: bool = False,
             want_terminal_obs: bool = False, sanitize_late: bool = False):
        """One transition of all N envs.  Returns (obs, reward, done, mask): views of buffers that
        the next step overwrites.  sanitize_late (Env_3 without masking): validate the press action after
        sort_material, as the reference's mode='random' does (MSE_STEP_SANITIZE_LATE)."""
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if actions.numel() != self.num_envs:
            raise ValueError("actions must have one entry per env")
        if sort_mode is not None:
            sort_mode = sort_mode.to(device=self.device, dtype=torch.int32).contiguous()
        flags = (0 if use_action_masking else MSE_STEP_UNMASKED) | (MSE_STEP_CHECK_OVERFLOW if check_overflow else 0) | \
                (MSE_STEP_SANITIZE_LATE if sanitize_late else 0)
        with torch.cuda.device(self.device):
            check(self.L.mse_step(self._h, _ptr(actions), _ptr(sort_mode), flags, _ptr(self.obs), _ptr(self.reward),
                                  _ptr(self.reward64) if want_reward64 else None, _ptr(self.done), _ptr(self.mask),
                                  _ptr(self.terminal_obs) if want_terminal_obs else None, self._stream()))
        if self._trace_buf is not None:
            self._trace_n += 1
        return self.obs, self.reward, self.done, self.mask

    def action_masks(self) -> torch.Tensor:
        out = torch.empty((self.num_envs, self.num_actions), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.mse_action_masks(self._h, _ptr(out), self._stream()))
        return out

    def sort_agent_obs(self) -> torch.Tensor:
        """f32[N, 13]: what Env_2_Pressing.step hands its sorting agent on the coming step (env_2_press.py:101):
        get_sort_obs() after that step's flow update.  Feed the agent's decisions back as step(..., sort_mode=)."""
        out = torch.empty((self.num_envs, 13), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.mse_sort_agent_obs(self._h, _ptr(out), self._stream()))
        return out

    def press_agent_obs(self) -> torch.Tensor:
        """f32[N, 16]: get_press_obs() after the coming step's flow update - what Env_3_Monolith.step(mode='model')
        hands its press_agent (env_monolith.py:198-210)."""
        out = torch.empty((self.num_envs, 16), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.mse_press_agent_obs(self._h, _ptr(out), self._stream()))
        return out

    def sample_actions(self, policy_seed: int = 2024) -> torch.Tensor:
        out = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.mse_sample_actions(self._h, int(policy_seed), _ptr(out), self._stream()))
        return out

    def rule_actions(self) -> torch.Tensor:
        """The reference's rule-based policy (mode='rule_based') evaluated on the device."""
        out = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.mse_rule_actions(self._h, _ptr(out), self._stream()))
        return out

    def model_actions(self, use_action_masking: bool = True, draw_sort: bool = True, draw_press: bool = True) -> torch.Tensor:
        """Env_3_Monolith.step(mode='model') with no agents assigned (env_monolith.py:186-221): the env's own
        rng_sorting / rng_pressing draws, NumPy-exact; the streams drawn from advance (draw_sort / draw_press = False:
        an assigned agent decides that part, it comes back as 0).  Step the result with masking semantics."""
        out = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        flags = (0 if use_action_masking else MSE_STEP_UNMASKED) | (0 if draw_sort else MSE_MODEL_NO_SORT_DRAW) | \
                (0 if draw_press else MSE_MODEL_NO_PRESS_DRAW)
        with torch.cuda.device(self.device):
            check(self.L.mse_model_actions(self._h, flags, _ptr(out), self._stream()))
        return out

    # ---- opt-in trace of one env (the reference's dashboard ledgers) ---------------------------------
    def trace_begin(self, env_index: int = 0, capacity: int = 4096) -> None:
        """From now on every step() appends one record f64[MSE_TRACE_COLS] for env `env_index` (mse_trace_begin)."""
        self._trace_buf = torch.zeros((int(capacity), MSE_TRACE_COLS), dtype=torch.float64, device=self.device)
        self._trace_n = 0
        check(self.L.mse_trace_begin(self._h, int(env_index), _ptr(self._trace_buf), int(capacity)))

    def trace_records(self) -> torch.Tensor:
        """The records written so far (synchronises the stream); the trace stays attached."""
        if getattr(self, "_trace_buf", None) is None:
            return torch.zeros((0, MSE_TRACE_COLS), dtype=torch.float64, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        return self._trace_buf[: self._trace_n]

    def trace_end(self) -> torch.Tensor:
        """Detaches the trace and returns its records f64[n, MSE_TRACE_COLS]."""
        rec = self.trace_records().clone()
        check(self.L.mse_trace_end(self._h, None))
        self._trace_buf = None
        return rec

    # ---- policy-stream counter (host-side handle state, see include/mse.h) ---------------------------
    @property
    def policy_step(self) -> int:
        v = C.c_uint64(0)
        check(self.L.mse_get_policy_step(self._h, C.byref(v)))
        return int(v.value)

    @policy_step.setter
    def policy_step(self, t: int) -> None:
        check(self.L.mse_set_policy_step(self._h, int(t)))

    def alloc_rollout(self, k_steps: int, obs=True, mask=True, actions=True, reward=True, done=True):
        n, dev, K = self.num_envs, self.device, int(k_steps)
        return {
            "actions": torch.empty((K, n), dtype=torch.int32, device=dev) if actions else None,
            "obs": torch.empty((K, n, self.obs_dim), dtype=torch.float32, device=dev) if obs else None,
            "reward": torch.empty((K, n), dtype=torch.float32, device=dev) if reward else None,
            "done": torch.empty((K, n), dtype=torch.uint8, device=dev) if done else None,
            "mask": torch.empty((K, n, self.num_actions), dtype=torch.uint8, device=dev) if mask else None,
        }

    def rollout(self, k_steps: int, policy_seed: int = 2024, buffers: Optional[dict] = None,
                sort_mode: Optional[torch.Tensor] = None, use_action_masking: bool = True,
                check_overflow: bool = False, policy: str = "random") -> dict:
        """K fused steps in one kernel launch under an on-device policy: "random" (masked-uniform) or
        "rule_based" (the reference's mode='rule_based')."""
        if buffers is None:
            buffers = self.alloc_rollout(k_steps)
        if sort_mode is not None:
            sort_mode = sort_mode.to(device=self.device, dtype=torch.int32).contiguous()
        flags = (0 if use_action_masking else MSE_STEP_UNMASKED) | (MSE_STEP_CHECK_OVERFLOW if check_overflow else 0)
        if not use_action_masking and self.kind == "mono" and policy == "random":
            flags |= MSE_STEP_SANITIZE_LATE  # the reference's mode='random' sequencing (env_monolith.py:245-253)
        if policy == "rule_based":
            flags |= MSE_ROLLOUT_RULE_BASED
        elif policy != "random":
            raise ValueError("policy must be 'random' or 'rule_based'")
        with torch.cuda.device(self.device):
            check(self.L.mse_rollout(self._h, int(k_steps), int(policy_seed), _ptr(sort_mode), flags,
                                     _ptr(buffers.get("actions")), _ptr(buffers.get("obs")),
                                     _ptr(buffers.get("reward")), _ptr(buffers.get("done")),
                                     _ptr(buffers.get("mask")), self._stream()))
        return buffers

    # ---- state export / import (tests, checkpoint, dashboard trace) ---------------------------
    def get_state(self):
        n, dev = self.num_envs, self.device
        ints = torch.empty((n, MSE_SNAP_INTS), dtype=torch.int64, device=dev)
        dbls = torch.empty((n, 4), dtype=torch.float64, device=dev)
        rng = torch.empty((n, MSE_SNAP_RNG_WORDS), dtype=torch.int64, device=dev)  # raw u64 words
        with torch.cuda.device(self.device):
            check(self.L.mse_get_state(self._h, _ptr(ints), _ptr(dbls), _ptr(rng), self._stream()))
        return ints, dbls, rng

    def set_state(self, ints=None, dbls=None, rng=None):
        def prep(t, dt):
            return None if t is None else t.to(device=self.device, dtype=dt).contiguous()
        ints, dbls, rng = prep(ints, torch.int64), prep(dbls, torch.float64), prep(rng, torch.int64)
        with torch.cuda.device(self.device):
            check(self.L.mse_set_state(self._h, _ptr(ints), _ptr(dbls), _ptr(rng), self._stream()))

    def refresh_outputs(self):
        """Rewrites self.obs / self.mask from the current state (after set_state): a reset that resets no env."""
        none = torch.zeros((self.num_envs,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.mse_reset(self._h, None, _ptr(none), _ptr(self.obs), _ptr(self.mask), self._stream()))
        return self.obs, self.mask

    def error_count(self) -> int:
        v = C.c_uint64(0)
        check(self.L.mse_error_count(self._h, C.byref(v)))
        return int(v.value)

    @property
    def algorithmic_bytes_per_step(self) -> int:
        return int(self.L.mse_algorithmic_bytes_per_step(self._h))
