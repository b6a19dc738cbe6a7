"""Import the reference (read-only checkout) in the BUILD CONTAINER to pin the oracle.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (it does not exist on
the GPU box).  Used by oracle/gen_golden.py to generate tests/golden/*.npz and by
tests/test_oracle_vs_reference.py (skipped when the checkout is absent).

The reference imports three packages that are not installed here and that its step path never
computes with: `gymnasium` (base class + space objects), `seaborn` and `cv2` (plotting).  They are
replaced by inert placeholder modules below; no arithmetic of the path runs through them.
The reference reads `config.yml` relative to the CWD (src/envs_train/env_super.py:25,28), hence
the chdir.  Nothing of the reference is copied into this repository.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MSE_REFERENCE_ROOT", "/root/reference")
MATERIALS = ["A", "B", "C", "D", "E"]


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "envs_train", "env_super.py"))


def _install_placeholders():
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")

        class Env:  # minimal gymnasium.Env surface the reference touches
            @property
            def unwrapped(self):
                return self

            def close(self):
                pass

        class _Space:
            def seed(self, seed=None):
                return [seed]

        class Box(_Space):
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low = np.asarray(low, dtype=dtype)
                self.high = np.asarray(high, dtype=dtype)
                self.dtype = np.dtype(dtype)
                self.shape = self.low.shape

        class Discrete(_Space):
            def __init__(self, n):
                self.n = int(n)

        spaces = types.ModuleType("gymnasium.spaces")
        spaces.Box, spaces.Discrete = Box, Discrete
        gym.Env, gym.spaces = Env, spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    for name in ("seaborn", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


_classes = None


def load():
    """Returns {'sort': Env_1_Sorting, 'press': Env_2_Pressing, 'mono': Env_3_Monolith}."""
    global _classes
    if _classes is None:
        if not available():
            raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
        _install_placeholders()
        import matplotlib

        matplotlib.use("Agg")
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        os.chdir(REFERENCE_ROOT)
        from src.envs_train.env_1_sort import Env_1_Sorting
        from src.envs_train.env_2_press import Env_2_Pressing
        from src.envs_train.env_monolith import Env_3_Monolith

        _classes = {"sort": Env_1_Sorting, "press": Env_2_Pressing, "mono": Env_3_Monolith}
    return _classes


class ScriptedSortAgent:
    """Stands where the reference expects a trained sorting agent (Env_2.set_agents):
    predict() returns the next scripted mode."""

    def __init__(self):
        self.next_mode = 0
        self.last_obs = None  # what the reference handed to predict(): get_sort_obs() after the step's flow update

    def predict(self, obs, deterministic=True):
        self.last_obs = np.asarray(obs, dtype=np.float32).copy()
        return int(self.next_mode), None


class ScriptedPressAgent:
    """Stands where Env_3_Monolith.step(mode='model') expects a trained pressing agent (env_monolith.py:198-210).
    predict() records what it was handed - the observation and, if the env passed one, the action mask - and returns
    `decide(mask)`: the next scripted press action."""

    def __init__(self, decide):
        self.decide = decide
        self.last_obs = None
        self.last_mask = None

    def predict(self, obs, deterministic=True, action_masks=None):
        self.last_obs = np.asarray(obs, dtype=np.float32).copy()
        self.last_mask = None if action_masks is None else np.asarray(action_masks, dtype=np.uint8).copy()
        return int(self.decide(self.last_mask)), None


class MaskableScriptedPressAgent(ScriptedPressAgent):
    """The same agent in the shape the reference's `is_maskable` test looks for (env_monolith.py:201: a `policy`
    attribute and 'Maskable' in the type's name): the env then passes press_action_masks() to predict()."""
    policy = object()


class ScriptedMonoAgent(ScriptedPressAgent):
    """Stands where Env_3_Monolith.step expects a stored monolith agent (env_monolith.py:144-150): predict(obs,
    deterministic=True, action_masks=mask) -> flat action."""


def _rng_words(gen) -> list[int]:
    st = gen.bit_generator.state
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    m = (1 << 64) - 1
    return [s >> 64, s & m, inc >> 64, inc & m, int(st["has_uint32"]), int(st["uinteger"])]


def snapshot(env):
    """Same layout as orc_env_snapshot (oracle/oracle.py SNAP); fields the reference does not
    have (draws, episode) are -1."""
    from oracle.oracle import SNAP, SNAP_DBLS, SNAP_INTS  # layout only

    I = np.zeros(SNAP_INTS, dtype=np.int64)
    D = np.zeros(SNAP_DBLS, dtype=np.float64)
    I[SNAP["input"]] = env.current_material_input
    I[SNAP["belt"]] = env.current_material_belt
    I[SNAP["sorting"]] = env.current_material_sorting
    cm = env.container_materials
    I[SNAP["cont_true"]] = [int(cm[m]) for m in "ABCD"]
    I[SNAP["cont_false"]] = [int(cm[m + "_False"]) for m in "ABCD"]
    I[SNAP["cont_e"]] = int(cm["E"])
    ps = env.press_state
    for p in (1, 2):
        I[SNAP["press_timer"].start + p - 1] = int(ps[f"press_{p}"])
        mat = ps[f"material_{p}"]
        I[SNAP["press_mat"].start + p - 1] = MATERIALS.index(mat) if isinstance(mat, str) else -1
        I[SNAP["press_n"].start + p - 1] = int(ps[f"n_{p}"])
        I[SNAP["press_q100"].start + p - 1] = int(np.rint(float(ps[f"q_{p}"]) * 100.0))
    I[SNAP["mode"]] = int(env.sensor_current_setting)
    I[SNAP["last_press_started"]] = int(bool(env._last_press_started))
    I[SNAP["last_press_amount"]] = int(env._last_press_amount)
    I[SNAP["current_step"]] = int(env.current_step)
    g = env.input_generator
    I[SNAP["gen_first"]] = int(g.pattern_sequence[0])
    I[SNAP["gen_idx"]] = int(g.current_pattern_idx)
    I[SNAP["gen_counter"]] = int(g.step_counter)
    for k, m in enumerate(MATERIALS):
        bales = env.bale_count[m]
        I[SNAP["bale_count"].start + k] = len(bales)
        I[SNAP["bale_sum"].start + k] = sum(int(b[0]) for b in bales)
        I[SNAP["bale_last_size"].start + k] = int(bales[-1][0]) if bales else 0
        I[SNAP["bale_last_q"].start + k] = int(bales[-1][1]) if bales else 0
    code, mat = -1, -1
    if env.press_actions_per_timestep:
        c, m = env.press_actions_per_timestep[-1]
        code = int(c)
        if isinstance(m, str):
            mat = MATERIALS.index(m)
        elif m is None:
            mat = -1
        else:
            mat = int(m)
    I[SNAP["log_code"]] = code
    I[SNAP["log_mat"]] = mat
    ipa = 0
    if env.name == "sort" and code in (1, 2):
        ipa = (code - 1) * 5 + mat + 1
    I[SNAP["internal_press_action"]] = ipa
    I[SNAP["draws"]] = -1
    I[SNAP["episode"]] = -1
    D[0:4] = env.accuracy_belt
    D[4:8] = env.accuracy_sorter
    R = np.array(_rng_words(env.rng) + _rng_words(env.rng_noise) + _rng_words(env.rng_pressing) +
                 _rng_words(env.rng_sorting) + _rng_words(env.input_generator.rng), dtype=np.uint64)
    return I, D, R


def ledgers(env):
    """The reference's per-env Python ledgers as plain arrays (fixture data for the opt-in trace):
    reward_data (env_super.py:402-408, 928-946), press_actions_per_timestep (:631-637, 730-736) with material
    names / ids / None as 0..4 / -1, and the full bale_count lists (:661-687)."""
    rd = env.reward_data
    out = {
        "reward": np.asarray(rd["Reward"], dtype=np.float64).reshape(-1, 2),
        "total": np.asarray(rd.get("Total", []), dtype=np.float64),
        "setting": np.asarray(rd["Setting"], dtype=np.int64),
        "belt_occupancy": np.asarray(rd["Belt_Occupancy"], dtype=np.float64),
        "belt_proportions": np.asarray([[float(p[m]) for m in "ABCD"] for p in rd["Belt_Proportions"]], dtype=np.float64).reshape(-1, 4),
        "true": np.asarray([rd[f"{m}_True"] for m in MATERIALS], dtype=np.int64).T.reshape(-1, 5),
        "false": np.asarray([rd[f"{m}_False"] for m in MATERIALS], dtype=np.int64).T.reshape(-1, 5),
    }
    log = []
    for code, mat in env.press_actions_per_timestep:
        if isinstance(mat, str):
            mat = MATERIALS.index(mat)
        elif mat is None:
            mat = -1
        log.append((int(code), int(mat)))
    out["press_log"] = np.asarray(log, dtype=np.int64).reshape(-1, 2)
    for m in MATERIALS:
        out[f"bales_{m}"] = np.asarray([(int(a), int(b)) for a, b in env.bale_count[m]], dtype=np.int64).reshape(-1, 2)
    return out
