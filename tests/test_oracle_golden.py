"""The CPU oracle against the golden vectors generated from the imported reference."""
import os

import pytest

from oracle.oracle import OracleEnv, STEP_CHECK_OVERFLOW, STEP_UNMASKED
from tests import replay


class OracleDriver:
    def __init__(self, meta):
        self.env = OracleEnv(kind=meta["kind"], max_steps=meta["max_steps"], seed=meta["ctor_seed"],
                             noise_sorting=meta["noise_sorting"], balesize=meta["balesize"],
                             cfg=replay.oracle_config(meta))

    def reset(self, seed):
        return self.env.reset(seed)

    def step(self, action, sort_mode, flags):
        return self.env.step(action, sort_mode, use_action_masking=not (flags & STEP_UNMASKED),
                             check_overflow=bool(flags & STEP_CHECK_OVERFLOW))

    def action_masks(self):
        return self.env.action_masks()

    def snapshot(self):
        return self.env.snapshot()

    def sort_agent_obs(self):
        return self.env.sort_agent_obs()


@pytest.mark.parametrize("path", replay.fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(path):
    n = replay.replay(path, OracleDriver, reward_tol=1e-12)
    assert n > 0


def test_fixture_inventory():
    names = [os.path.basename(p) for p in replay.fixtures()]
    assert len(names) >= 30
    for kind in ("mono", "press", "sort"):
        assert any(n.startswith(kind + "_") for n in names)
