"""The CPU oracle against the fixtures of the reference's in-env action sources and per-env ledgers:
  model_*     Env_3_Monolith.step(mode='model') without agents (env_monolith.py:186-221): rng_sorting / rng_pressing draws
  randmode_*  Env_3_Monolith.step(mode='random') (env_monolith.py:152-164, 245-253): late sanitising without masking
  trace_*     reward_data / press_actions_per_timestep / bale_count at the end of an episode
All generated from the imported reference by oracle/gen_golden.py (--model / --random-mode / --trace)."""
import glob
import os

import numpy as np
import pytest

import marl_sortingenv_amd  # noqa: F401  (registers the package alias)
from marl_sortingenv_amd.trace import EnvTrace
from oracle.oracle import OracleEnv, STEP_CHECK_OVERFLOW, STEP_UNMASKED
from tests import replay


def paths(prefix):
    return sorted(glob.glob(os.path.join(replay.GOLDEN_DIR, prefix + "_*.npz")))


class ModeDriver:
    """Replays a model_* / randmode_* fixture: the step's action is produced (model) or given (random) and the
    flags follow from the mode, as the reference's step() does internally."""

    def __init__(self, meta):
        self.meta = meta
        self.mode = "model" if meta["name"].startswith("model") else "random"
        self.env = OracleEnv(kind="mono", max_steps=meta["max_steps"], seed=meta["ctor_seed"],
                             noise_sorting=meta["noise_sorting"], balesize=meta["balesize"])

    def reset(self, seed):
        return self.env.reset(seed)

    def step(self, action, sort_mode, flags):
        masking = self.meta["masking"]
        if self.mode == "model":
            drawn = self.env.model_fallback_action(masking)
            assert drawn == action, f"mode='model' drew {drawn}, the reference {action}"
            return self.env.step(action)                      # applied unsanitised (env_monolith.py:254-257)
        return self.env.step(action, use_action_masking=masking, sanitize_late=not masking)

    def action_masks(self):
        return self.env.action_masks()

    def snapshot(self):
        return self.env.snapshot()


@pytest.mark.parametrize("path", paths("model") + paths("randmode"), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_in_env_action_sources(path):
    assert replay.replay(path, ModeDriver, reward_tol=1e-12) > 400


def compare_ledgers(z, got, tol):
    for key in ("setting", "true", "false", "press_log") + tuple(f"bales_{m}" for m in "ABCDE"):
        exp = z["ledger_" + key]
        assert exp.shape == got[key].shape and np.array_equal(exp, got[key]), key
    for key in ("reward", "total", "belt_occupancy", "belt_proportions"):
        exp = z["ledger_" + key]
        assert exp.shape == got[key].shape and np.all(np.abs(exp - got[key]) <= tol), key


@pytest.mark.parametrize("path", paths("trace") + paths("randmode"), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_ledgers(path):
    """The oracle's per-step records assembled by the product's EnvTrace equal the reference's ledgers."""
    meta, z = replay.load(path)
    env = OracleEnv(kind=meta["kind"], max_steps=meta["max_steps"], seed=meta["ctor_seed"],
                    noise_sorting=meta["noise_sorting"], balesize=meta["balesize"])
    tr = EnvTrace(meta["balesize"], 0.5)
    late = meta["name"].startswith("randmode") and not meta["masking"]
    for t in range(len(z["op"])):
        if z["op"][t] == 1:
            env.reset(int(z["arg"][t]))
            tr.reset()
        else:
            f = int(z["flags"][t])
            masking = meta["masking"] if meta["name"].startswith("randmode") else not (f & STEP_UNMASKED)
            env.step(int(z["arg"][t]), int(z["sort_mode"][t]), use_action_masking=masking,
                     check_overflow=bool(f & STEP_CHECK_OVERFLOW), sanitize_late=late)
            tr.append(env.trace_record())
    compare_ledgers(z, tr.as_arrays(), 1e-12)
    assert tr.bale_count == env.bale_count()
