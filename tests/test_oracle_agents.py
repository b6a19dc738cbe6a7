"""The CPU oracle against the fixtures of Env_3_Monolith.step() with AGENTS assigned and of the overflow `info` keys
(oracle/gen_golden.py --agents / --overflow-info, from the imported reference):
  agents_mono_monoagent_*  the observation / mask a stored mono_agent is handed inside step() (env_monolith.py:144-150)
  agents_mono_model_*      mode='model' with a sort_agent and / or a press_agent (env_monolith.py:186-221)
  info_*                   {"overflow", "overflow_material", "action"} of a check_overflow termination
                           (env_monolith.py:264-268, env_2_press.py:144-150, env_1_sort.py:131-139)."""
import os

import numpy as np
import pytest

from oracle.oracle import OracleEnv
from tests import replay
from tests.test_oracle_modes_trace import paths


# the input generator's private stream is only followed when the batch size leaves a floor() remainder
# (tests/test_oracle_golden.py does the same for the remainder-free step traces)
GENERATOR_WORDS = tuple(range(24, 30))


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


class AgentsDriver:
    """Replays an agents_* fixture on a driver `env` offering the oracle's surface: before each step the previews an
    agent would be handed are compared with what the reference handed its agents, the env's own draws fill the parts
    no agent decides, and the action is applied unsanitised (env_monolith.py:254-257)."""
    make_env = staticmethod(lambda meta: OracleEnv(kind="mono", max_steps=meta["max_steps"], seed=meta["ctor_seed"],
                                                   noise_sorting=meta["noise_sorting"], balesize=meta["balesize"]))

    def __init__(self, meta, z):
        self.meta, self.z, self.t = meta, z, 0
        self.env = self.make_env(meta)

    def reset(self, seed):
        self.t += 1
        return self.env.reset(seed)

    def previews(self):
        return self.env.sort_agent_obs(), self.env.press_agent_obs(), self.env.action_masks()

    def model_action(self, masking, draw_sort, draw_press):
        return self.env.model_action(masking, draw_sort, draw_press)

    def apply(self, action):
        return self.env.step(action)

    def step(self, action, sort_mode, flags):
        z, t, ag = self.z, self.t, self.meta["agents"]
        so, po, mask = self.previews()
        if ag["mono"]:
            assert np.array_equal(bits(np.concatenate([so, po])), bits(z["agent_obs"][t])), f"row {t}: mono_agent's observation"
            assert np.array_equal(np.asarray(mask, dtype=np.uint8), z["agent_mask"][t]), f"row {t}: mono_agent's mask"
        else:
            if ag["sort"]:
                assert np.array_equal(bits(so), bits(z["sort_agent_obs"][t])), f"row {t}: sort_agent's observation"
            if ag["press"]:
                assert np.array_equal(bits(po), bits(z["press_agent_obs"][t])), f"row {t}: press_agent's observation"
                # the mask reaches a press agent only if it is maskable and masking is on (env_monolith.py:201-205)
                assert bool(z["press_agent_has_mask"][t]) == (ag["press"] == "maskable" and self.meta["masking"])
                if z["press_agent_has_mask"][t]:
                    assert np.array_equal(np.asarray(mask, dtype=np.uint8)[:11], z["press_agent_mask"][t]), f"row {t}: press mask"
            drawn = self.model_action(self.meta["masking"], not ag["sort"], ag["press"] is None)
            if not ag["sort"]:
                assert drawn // 11 == action // 11, f"row {t}: rng_sorting draw"
            if ag["press"] is None:
                assert drawn % 11 == action % 11, f"row {t}: rng_pressing draw"
        self.t += 1
        return self.apply(action)

    def action_masks(self):
        return self.env.action_masks()

    def snapshot(self):
        return self.env.snapshot()


def replay_agents(path, driver_cls, reward_tol, **kw):
    meta, z = replay.load(path)
    return replay.replay(path, lambda m: driver_cls(m, z), reward_tol=reward_tol, **kw)


@pytest.mark.parametrize("path", paths("agents"), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_agents(path):
    assert replay_agents(path, AgentsDriver, 1e-12, skip_rng_words=GENERATOR_WORDS) > 400


def info_of(rec, kind, action):
    """The reference's info dict from a trace record (include/mse.h MSE_TRACE_*), as (overflow, material, action)."""
    ovf = int(rec[37])
    if ovf and kind == "sort":
        return 1, ovf - 1, int(rec[31])           # env_1_sort.py:139: the press job the env sampled itself
    return int(ovf > 0), ovf - 1 if ovf else -1, int(action)


@pytest.mark.parametrize("path", paths("info"), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_overflow_info(path):
    meta, z = replay.load(path)
    ends = []

    class Driver:
        def __init__(self, m):
            self.env = OracleEnv(kind=m["kind"], cfg=replay.oracle_config(m), seed=m["ctor_seed"])
            self.t = 0

        def reset(self, seed):
            self.t += 1
            return self.env.reset(seed)

        def step(self, action, sort_mode, flags):
            out = self.env.step(action, sort_mode, check_overflow=True)
            got = info_of(self.env.trace_record(), meta["kind"], action)
            exp = (int(z["info_overflow"][self.t]), int(z["info_material"][self.t]), int(z["info_action"][self.t]))
            assert got == exp, f"row {self.t}: info {got} vs the reference's {exp}"
            if out[2]:
                ends.append(got[1])
            self.t += 1
            return out

        def action_masks(self):
            return self.env.action_masks()

        def snapshot(self):
            return self.env.snapshot()

    replay.replay(path, Driver, reward_tol=1e-12, skip_rng_words=GENERATOR_WORDS)
    assert len(ends) >= 2
