"""ctypes binding of libmse_hip.so (include/mse.h).  No CPU fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

MSE_ENV_SORT, MSE_ENV_PRESS, MSE_ENV_MONO = 1, 2, 3
MSE_STEP_UNMASKED, MSE_STEP_CHECK_OVERFLOW, MSE_ROLLOUT_RULE_BASED, MSE_STEP_SANITIZE_LATE = 1, 2, 4, 8
MSE_MODEL_NO_SORT_DRAW, MSE_MODEL_NO_PRESS_DRAW = 16, 32
MSE_SNAP_INTS = 71
MSE_SNAP_RNG_WORDS = 30  # rng, rng_noise, rng_pressing, rng_sorting, input generator x {state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger}
MSE_TRACE_COLS = 40

EXPORTS = [
    "mse_version", "mse_last_error", "mse_status_string", "mse_config_default", "mse_create",
    "mse_create_indexed", "mse_destroy", "mse_num_envs", "mse_obs_dim", "mse_num_actions", "mse_reset",
    "mse_step", "mse_action_masks", "mse_rollout", "mse_sample_actions", "mse_rule_actions", "mse_get_state", "mse_set_state",
    "mse_error_count", "mse_algorithmic_bytes_per_step", "mse_tie_window",
    "mse_sort_agent_obs", "mse_policy_num_weights", "mse_policy_create", "mse_policy_destroy", "mse_policy_forward",
    "mse_get_policy_step", "mse_set_policy_step", "mse_model_actions", "mse_trace_begin", "mse_trace_end", "mse_press_agent_obs", "mse_rollout_policy", "mse_policy_set_precision", "mse_policy_precision",
]

_other_libs: dict = {}


class MseError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libmse_hip status {status}: {message}")
        self.status = status


class MseConfigStruct(C.Structure):
    """struct mse_config (include/mse.h)"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("env_kind", C.c_int32), ("max_steps", C.c_int32),
        ("auto_reset", C.c_int32), ("track_bales", C.c_int32), ("literal_choice", C.c_int32),
        ("input_batch_size", C.c_int32), ("steps_per_pattern", C.c_int32),
        ("baseline_accuracy", C.c_double * 4), ("boost", C.c_double), ("noise", C.c_double),
        ("stage_capacity", C.c_int32),
        ("press_time", C.c_int32 * 2), ("container_capacity", C.c_int32), ("bale_standard_size", C.c_int32),
        ("bale_remainder_threshold", C.c_double), ("quality_threshold", C.c_double * 4),
        ("quality_threshold_r2", C.c_double * 4),
        ("purity_threshold_theta", C.c_double), ("tanh_temperature", C.c_double),
        ("overflow_penalty_catastrophic", C.c_double), ("overflow_penalty_severe", C.c_double),
        ("overflow_penalty_mild", C.c_double), ("bale_efficiency_factor", C.c_double),
        ("max_state_reward", C.c_double), ("overflow_termination_penalty", C.c_double),
        ("pattern_ratio", (C.c_double * 4) * 2),
        ("rollout_pipeline", C.c_int32), ("reserved0", C.c_int32),
    ]


_lib = None


def library_path() -> str:
    return LIB_PATH


def load_library(path: str | None = None) -> C.CDLL:
    """Loads the in-tree libmse_hip.so; raises if it has not been built (python -m ... build).
    `path` names another build of the same source (tests load libmse_hip_widetie.so beside the product)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and path in _other_libs:
        return _other_libs[path]
    lib_path = LIB_PATH if path is None else path
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: build it with `python __graft_entry__.py` or "
            "`python marl-sortingenv_amd/build.py`. There is no CPU fallback for the step path.")
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as ROCm's). Import it first so that
    # libmse_hip.so binds to the runtime torch uses: streams and device pointers are shared objects.
    import torch  # noqa: F401

    L = C.CDLL(lib_path)
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    L.mse_version.restype = C.c_int
    L.mse_last_error.restype = C.c_char_p
    L.mse_status_string.argtypes = [C.c_int]
    L.mse_status_string.restype = C.c_char_p
    L.mse_config_default.argtypes = [C.POINTER(MseConfigStruct)]
    L.mse_create.argtypes = [C.POINTER(vp), C.POINTER(MseConfigStruct), i64, C.c_int]
    L.mse_create_indexed.argtypes = [C.POINTER(vp), C.POINTER(MseConfigStruct), i64, C.c_int, i64]
    L.mse_destroy.argtypes = [vp]
    L.mse_num_envs.argtypes = [vp]
    L.mse_num_envs.restype = i64
    L.mse_obs_dim.argtypes = [vp]
    L.mse_num_actions.argtypes = [vp]
    L.mse_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    L.mse_step.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.mse_action_masks.argtypes = [vp, vp, vp]
    L.mse_sort_agent_obs.argtypes = [vp, vp, vp]
    L.mse_press_agent_obs.argtypes = [vp, vp, vp]
    L.mse_rollout.argtypes = [vp, i32, u64, vp, u32, vp, vp, vp, vp, vp, vp]
    L.mse_sample_actions.argtypes = [vp, u64, vp, vp]
    L.mse_rule_actions.argtypes = [vp, vp, vp]
    L.mse_get_state.argtypes = [vp, vp, vp, vp, vp]
    L.mse_set_state.argtypes = [vp, vp, vp, vp, vp]
    L.mse_error_count.argtypes = [vp, C.POINTER(u64)]
    L.mse_algorithmic_bytes_per_step.argtypes = [vp]
    L.mse_get_policy_step.argtypes = [vp, C.POINTER(u64)]
    L.mse_set_policy_step.argtypes = [vp, u64]
    L.mse_model_actions.argtypes = [vp, u32, vp, vp]
    L.mse_trace_begin.argtypes = [vp, i64, vp, i64]
    L.mse_trace_end.argtypes = [vp, C.POINTER(i64)]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError if the library does not export what include/mse.h declares
    L.mse_tie_window.restype = u32
    L.mse_policy_num_weights.argtypes = [C.c_int, C.c_int]
    L.mse_policy_num_weights.restype = i64
    L.mse_policy_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
    L.mse_policy_destroy.argtypes = [vp]
    L.mse_policy_set_precision.argtypes = [vp, C.c_int]
    L.mse_policy_precision.argtypes = [vp]
    L.mse_policy_forward.argtypes = [vp, i64, i64, vp, vp, u64, u64, C.c_int, vp, vp, vp, vp, vp]
    L.mse_rollout_policy.argtypes = [vp, vp, vp, i32, u64, C.c_int, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    if path is None:
        _lib = L
    else:
        _other_libs[path] = L
    return L


def check(status: int):
    if status != 0:
        raise MseError(status, load_library().mse_last_error().decode(errors="replace"))
