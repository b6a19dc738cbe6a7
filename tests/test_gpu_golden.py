"""PARITY (GPU): the HIP step engine, through the C ABI, against the golden vectors generated from
the imported reference.  Integer state, masks, PCG64 states and f32 observations bit for bit;
rewards within 1e-6 (north_star tolerance; the only inexact op is fp64 tanh)."""
import os

import numpy as np
import pytest

from oracle.oracle import SNAP, STEP_CHECK_OVERFLOW, STEP_UNMASKED
from tests import replay

pytestmark = pytest.mark.gpu

# per-lane state the engine deliberately does not carry (DESIGN.md "what is not carried per lane")
SKIP_COLS = [SNAP["log_code"].start, SNAP["log_mat"].start, SNAP["internal_press_action"].start]
# the noise stream is not advanced when noise == 0 (unobservable); has_uint32/uinteger are only
# meaningful for rng_pressing
NOISE_WORDS = list(range(6, 12))
ALWAYS_SKIP_WORDS = [4, 5, 10, 11]


class GpuDriver:
    def __init__(self, meta, literal_choice=False, library=None):
        import torch
        import marl_sortingenv_amd as M

        self.torch = torch
        self.env = M.BatchedSortingEnv(kind=meta["kind"], num_envs=1, device=0, base_seed=meta["ctor_seed"],
                                       max_steps=meta["max_steps"], noise_sorting=meta["noise_sorting"],
                                       balesize=meta["balesize"], auto_reset=False, literal_choice=literal_choice,
                                       library=library, config=replay.sorting_config(meta))
        self.prev_acc = None
        self.acc_sorter = None

    def _acc(self):
        return self.env.get_state()[1][0].cpu().numpy()

    def reset(self, seed):
        t = self.torch
        obs, _ = self.env.reset(seeds=t.tensor([seed], dtype=t.int64))
        self.acc_sorter = self._acc().copy()
        return obs[0].cpu().numpy()

    def step(self, action, sort_mode, flags):
        t = self.torch
        before = self._acc().copy()
        sm = None if sort_mode < 0 else t.tensor([sort_mode], dtype=t.int32)
        obs, rew, done, _ = self.env.step(t.tensor([action], dtype=t.int32), sort_mode=sm,
                                          use_action_masking=not (flags & STEP_UNMASKED),
                                          check_overflow=bool(flags & STEP_CHECK_OVERFLOW), want_reward64=True)
        self.acc_sorter = before  # accuracy_sorter <- accuracy_belt at the start of the step
        return obs[0].cpu().numpy(), float(self.env.reward64[0].item()), bool(done[0].item())

    def action_masks(self):
        return self.env.action_masks()[0].cpu().numpy()

    def sort_agent_obs(self):
        return self.env.sort_agent_obs()[0].cpu().numpy()

    def snapshot(self):
        ints, dbls, rng = self.env.get_state()
        d = np.concatenate([dbls[0].cpu().numpy(), self.acc_sorter])
        return ints[0].cpu().numpy(), d, rng[0].cpu().numpy().view(np.uint64)


GENERATOR_WORDS = list(range(24, 30))  # the input generator's private stream: only advanced (and compared) when the
                                       # fixture's batch size leaves a floor() remainder


def _skip_words(meta):
    batch = ((meta.get("config_overrides") or {}).get("simulation") or {}).get("input_batch_size", 100)
    remainder_free = all(sum(int(np.floor(r * batch)) for r in p) == batch
                         for p in ((0.40, 0.15, 0.35, 0.10), (0.15, 0.40, 0.10, 0.35)))
    return ALWAYS_SKIP_WORDS + (NOISE_WORDS if meta["noise_sorting"] == 0.0 else []) + (GENERATOR_WORDS if remainder_free else [])


@pytest.mark.parametrize("path", replay.fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_hip_matches_golden(path):
    n = replay.replay(path, GpuDriver, reward_tol=1e-6, skip_cols=SKIP_COLS, skip_rng_words=_skip_words)
    assert n > 0


@pytest.mark.parametrize("name", ["mono_n5_masked_s0_s3", "press_n0_unmasked_s3_s4294967293", "sort_n5_masked_s1_s42"])
def test_hip_literal_choice_matches_golden(name):
    """Same vectors with Generator.choice evaluated in literal fp64 for every draw (the fallback
    path of the exact integer decision)."""
    path = os.path.join(replay.GOLDEN_DIR, name + ".npz")
    replay.replay(path, lambda meta: GpuDriver(meta, literal_choice=True), reward_tol=1e-6,
                  skip_cols=SKIP_COLS, skip_rng_words=_skip_words)
