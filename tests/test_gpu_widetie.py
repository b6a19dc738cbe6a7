"""PARITY (GPU): the literal-cdf branches of the `choice` decision under load.

With the shipped window a draw falls back to the literal fp64 cdf about once in 10^7 draws, so the
ordinary parity tests almost never execute those branches (nor the ring kernel's redo of a whole
step).  libmse_hip_widetie.so is the same source built with MSE_TIE_WINDOW = 2^27: ~3 % of the
draws (about every third step) take the fallback, and the results must still be the golden ones."""
import os

import pytest

from tests import replay
from tests.test_gpu_golden import SKIP_COLS, GpuDriver, _skip_words

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDE = os.path.join(ROOT, "marl-sortingenv_amd", "libmse_hip_widetie.so")


def _wide_lib():
    import marl_sortingenv_amd as M

    assert os.path.exists(WIDE), "build libmse_hip_widetie.so first: python __graft_entry__.py"
    L = M.load_library(WIDE)
    assert L.mse_tie_window() == 0x08000000
    assert M.load_library().mse_tie_window() == 162  # f < 2 or f >= 2^32 - 160 (csrc/mse_device.h)
    return WIDE


@pytest.mark.parametrize("path", replay.fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_wide_window_matches_golden(path):
    lib = _wide_lib()
    n = replay.replay(path, lambda meta: GpuDriver(meta, library=lib), reward_tol=1e-6, skip_cols=SKIP_COLS,
                      skip_rng_words=_skip_words)
    assert n > 0


@pytest.mark.parametrize("kind", ["mono", "press", "sort"])
@pytest.mark.parametrize("pipeline", [1, 2, 3])
def test_wide_window_rollout_kernels_equal_shipped_literal(kind, pipeline):
    """Every rollout kernel of the wide-window build (3 = ring: whole-step redo; 1, 2: per-draw fallback)
    against the shipped library with literal_choice (every draw by the fp64 cdf)."""
    import torch

    import marl_sortingenv_amd as M

    lib = _wide_lib()
    n, K = 3000, 48
    kw = dict(kind=kind, num_envs=n, device=0, base_seed=31, max_steps=40, noise_sorting=0.0, balesize=200)
    a = M.BatchedSortingEnv(rollout_pipeline=pipeline, library=lib, **kw)
    b = M.BatchedSortingEnv(rollout_pipeline=2, literal_choice=True, **kw)
    ra, rb = a.rollout(K, policy_seed=5), b.rollout(K, policy_seed=5)
    for key in ra:
        assert torch.equal(ra[key], rb[key]), key
    for x, y in zip(a.get_state(), b.get_state()):
        assert torch.equal(x, y)
    assert a.error_count() == 0


@pytest.mark.parametrize("kind,n,pipeline", [("mono", 3000, 0), ("mono", 3000, 1), ("mono", 3000, 2), ("sort", 700, 0),
                                             ("press", 70300, 0)])
def test_wide_window_fused_policy_rollout_equals_shipped(kind, n, pipeline):
    """mse_rollout_policy steps the env with the windowed decision too: the wide-window build (fallback about
    every third step; every kernel shape: actor / critic / RNG waves with the ring's whole-step redo (0), actor / critic
    waves (1), plain 32-env waves (2), 64-env waves above 65 536 envs) must collect what the shipped build collects."""
    import torch

    import marl_sortingenv_amd as M
    from tests.test_gpu_policy import _weights

    lib = _wide_lib()
    K = 20
    kw = dict(kind=kind, num_envs=n, device=0, base_seed=23, max_steps=15, noise_sorting=0.05, balesize=200)
    a, b = M.BatchedSortingEnv(library=lib, rollout_pipeline=pipeline, **kw), M.BatchedSortingEnv(**kw)
    w = _weights(a.obs_dim, a.num_actions, seed=9)
    fa = M.FusedPolicyRollout(a, M.MlpPolicy(a.obs_dim, a.num_actions, w, device=0, library=lib), K, seed=4)
    fb = M.FusedPolicyRollout(b, M.MlpPolicy(b.obs_dim, b.num_actions, w, device=0), K, seed=4)
    for it in range(2):
        x, y = fa.collect(), fb.collect()
        for key in x:
            assert torch.equal(x[key], y[key]), (it, key)
    for sa, sb in zip(a.get_state(), b.get_state()):
        assert torch.equal(sa, sb)
    assert a.error_count() == 0
