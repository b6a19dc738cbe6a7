"""Oracle vs the imported reference under fuzzed config.yml's (build container only): pins the CPU restatement off the
three hand-written configurations of tests/golden/.  Every row: observation bits, reward (1e-12), mask, integer state,
accuracies and all five PCG64 streams (the generator's only where the batch size makes it observable)."""
import numpy as np
import pytest

from oracle import ref_harness
from oracle.oracle import OracleEnv
from tests import replay
from tests.fuzz_configs import fuzz_overrides, meta_for

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference checkout not present")


@pytest.mark.parametrize("seed", range(30))
def test_oracle_matches_reference_under_a_fuzzed_config(seed):
    from oracle.gen_golden import _construct

    kind = ("mono", "press", "sort")[seed % 3]
    masking = seed % 4 != 3 or kind == "sort"
    check_overflow = seed % 5 == 4   # overflow terminates with the penalty (default off in the reference's callers)
    hoard = seed % 6 == 5            # mostly no-ops: containers far beyond their capacity
    meta = meta_for(kind, seed)
    ov, ctor = fuzz_overrides(seed)
    ref = _construct(ref_harness.load()[kind], seed, ctor, ov)
    agent = None
    if kind == "press":
        agent = ref_harness.ScriptedSortAgent()
        ref.set_agents(sort_agent=agent)
    orc = OracleEnv(kind=kind, seed=seed, cfg=replay.oracle_config(meta))
    batch = ov["simulation"]["input_batch_size"]
    gen_observable = any(sum(int(np.floor(x * batch)) for x in p) != batch
                         for p in ((0.40, 0.15, 0.35, 0.10), (0.15, 0.40, 0.10, 0.35)))
    prng = np.random.default_rng(seed * 7 + 1)
    for episode in range(2):
        s = seed * 10 + episode
        o_ref, _ = ref.reset(seed=s)
        assert np.array_equal(np.asarray(o_ref, np.float32).view(np.uint32), orc.reset(s).view(np.uint32))
        for t in range(ctor["max_steps"]):
            mask = np.asarray(ref.action_masks(), dtype=np.uint8)
            assert np.array_equal(mask, orc.action_masks()), (episode, t)
            a = int(prng.choice(np.flatnonzero(mask))) if (masking and prng.random() < 0.8) else int(prng.integers(0, mask.size))
            if hoard and prng.random() < 0.9:
                a = (a // 11) * 11 if kind == "mono" else (0 if kind == "press" else a)
            sm = int(prng.integers(0, 2))
            if agent is not None:
                agent.next_mode = sm
            o_ref, r_ref, term_ref, _, _ = ref.step(a, use_action_masking=masking, check_overflow=check_overflow)
            o_orc, r_orc, term_orc = orc.step(a, sm, use_action_masking=masking, check_overflow=check_overflow)
            assert np.array_equal(np.asarray(o_ref, np.float32).view(np.uint32), o_orc.view(np.uint32)), (episode, t)
            assert abs(r_ref - r_orc) <= 1e-12 and term_ref == term_orc, (episode, t, r_ref, r_orc)
            I, D, R = ref_harness.snapshot(ref)
            Io, Do, Ro = orc.snapshot()
            Io[59] = Io[60] = I[59] = I[60] = 0
            assert np.array_equal(I, Io), (episode, t, np.flatnonzero(I != Io))
            n_words = 30 if gen_observable else 24
            assert np.array_equal(D, Do) and np.array_equal(R[:n_words], Ro[:n_words]), (episode, t)
            if term_ref:
                break


@pytest.mark.parametrize("seed", range(6))
def test_oracle_model_mode_matches_reference_under_a_fuzzed_config(seed):
    """Env_3_Monolith.step(action=None, mode='model') without agents (env_monolith.py:186-221) under fuzzed configs:
    the env's own rng_sorting / rng_pressing draws, masked and unmasked."""
    from oracle.gen_golden import _construct

    masking = seed % 2 == 0
    meta = meta_for("mono", 40 + seed)
    ov, ctor = fuzz_overrides(40 + seed)
    ref = _construct(ref_harness.load()["mono"], seed, ctor, ov)
    orc = OracleEnv(kind="mono", seed=seed, cfg=replay.oracle_config(meta))
    o_ref, _ = ref.reset(seed=seed)
    assert np.array_equal(np.asarray(o_ref, np.float32).view(np.uint32), orc.reset(seed).view(np.uint32))
    for t in range(ctor["max_steps"]):
        o_ref, r_ref, term_ref, _, info = ref.step(action=None, mode="model", use_action_masking=masking)
        a = orc.model_fallback_action(masking)
        assert a == int(info["action"]), (t, a, info)
        o_orc, r_orc, term_orc = orc.step(a)
        assert np.array_equal(np.asarray(o_ref, np.float32).view(np.uint32), o_orc.view(np.uint32)), t
        assert abs(r_ref - r_orc) <= 1e-12 and term_ref == term_orc
        R, Ro = ref_harness.snapshot(ref)[2], orc.snapshot()[2]
        assert np.array_equal(R[:24], Ro[:24]), t
