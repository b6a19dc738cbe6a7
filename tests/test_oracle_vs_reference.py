"""Live check of the oracle against the imported reference on fresh random scripts (build
container only: skipped wherever the reference checkout is absent, e.g. on the GPU box)."""
import numpy as np
import pytest

from oracle import ref_harness
from oracle.oracle import OracleEnv

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference checkout not present")


@pytest.mark.parametrize("kind", ["mono", "press", "sort"])
@pytest.mark.parametrize("seed,noise,masking", [(11, 0.0, True), (2024, 0.05, True), (77, 0.02, False)])
def test_random_script(kind, seed, noise, masking):
    if kind == "sort" and not masking:
        pytest.skip("Env_1 ignores use_action_masking")
    cls = ref_harness.load()[kind]
    ref = cls(max_steps=60, seed=seed, noise_sorting=noise, balesize=180)
    agent = None
    if kind == "press":
        agent = ref_harness.ScriptedSortAgent()
        ref.set_agents(sort_agent=agent)
    orc = OracleEnv(kind=kind, max_steps=60, seed=seed, noise_sorting=noise, balesize=180)
    prng = np.random.default_rng(seed * 3 + 1)
    for episode in range(3):
        s = seed + episode
        o_ref, _ = ref.reset(seed=s)
        o_orc = orc.reset(s)
        assert np.array_equal(o_ref, o_orc)
        for t in range(60):
            mask = np.asarray(ref.action_masks(), dtype=np.uint8)
            assert np.array_equal(mask, orc.action_masks())
            if masking and prng.random() < 0.8:
                a = int(prng.choice(np.flatnonzero(mask)))
            else:
                a = int(prng.integers(0, mask.size))
            sm = int(prng.integers(0, 2))
            if agent is not None:
                agent.next_mode = sm
            o_ref, r_ref, term_ref, _, _ = ref.step(a, use_action_masking=masking)
            o_orc, r_orc, term_orc = orc.step(a, sm, use_action_masking=masking)
            assert np.array_equal(np.asarray(o_ref, np.float32).view(np.uint32), o_orc.view(np.uint32)), (episode, t)
            assert abs(r_ref - r_orc) <= 1e-12
            assert term_ref == term_orc
            I, D, R = ref_harness.snapshot(ref)
            Io, Do, Ro = orc.snapshot()
            Io[59] = Io[60] = I[59] = I[60] = 0
            assert np.array_equal(I, Io), (episode, t, np.flatnonzero(I != Io))
            # (the generator's private stream - words 24.. - is only followed when the batch size leaves a floor()
            # remainder: tests/golden/*_cfgB_*; with config.yml's 100 it is never observed)
            assert np.array_equal(D, Do) and np.array_equal(R[:24], Ro[:24])
