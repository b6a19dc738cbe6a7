"""MI355X-native batched step() engine for the MARL-SortingEnv environments.

Hot path only: the per-env state transition of Env_1_Sorting /
Env_2_Pressing / Env_3_Monolith as hand-written HIP kernels behind a C ABI (include/mse.h),
with the reference's Gymnasium reset()/step()/action_masks() surface on the host side.
"""
from ._lib import (MSE_ENV_MONO, MSE_ENV_PRESS, MSE_ENV_SORT, MSE_SNAP_INTS, MSE_STEP_CHECK_OVERFLOW,
                   MSE_STEP_UNMASKED, MseError, library_path, load_library)
from .build import build_library
from .config import KIND_BY_NAME, NUM_ACTIONS, OBS_DIM, SortingEnvConfig

__all__ = [
    "BatchedSortingEnv", "Env_1_Sorting", "Env_2_Pressing", "Env_3_Monolith", "SortingVecEnv",
    "ShardedSortingEnv", "MlpPolicy", "PolicyRolloutCollector", "FusedPolicyRollout", "SortingEnvConfig", "MseError", "build_library", "load_library", "library_path",
    "KIND_BY_NAME", "OBS_DIM", "NUM_ACTIONS", "MSE_ENV_SORT", "MSE_ENV_PRESS", "MSE_ENV_MONO",
    "MSE_STEP_UNMASKED", "MSE_STEP_CHECK_OVERFLOW", "MSE_SNAP_INTS",
]


def __getattr__(name):  # torch-dependent front-ends are imported on first use
    if name == "BatchedSortingEnv":
        from .batched import BatchedSortingEnv
        return BatchedSortingEnv
    if name in ("Env_1_Sorting", "Env_2_Pressing", "Env_3_Monolith"):
        from . import envs
        return getattr(envs, name)
    if name == "SortingVecEnv":
        from .vec_env import SortingVecEnv
        return SortingVecEnv
    if name in ("PolicyRolloutCollector", "FusedPolicyRollout"):
        from . import collector
        return getattr(collector, name)
    if name == "MlpPolicy":
        from .policy import MlpPolicy
        return MlpPolicy
    if name == "ShardedSortingEnv":
        from .sharding import ShardedSortingEnv
        return ShardedSortingEnv
    raise AttributeError(name)
