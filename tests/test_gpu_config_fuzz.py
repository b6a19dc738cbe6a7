"""PARITY (GPU): the HIP engine vs the oracle under fuzzed configurations (tests/fuzz_configs.py; the oracle is held to
the imported reference under the same fuzz in tests/test_oracle_config_fuzz.py): tables, thresholds, press times, batch
sizes with and without a floor() remainder, noise levels - 64 envs x ~2.5 episodes each, through mse_step with
auto-reset, and the fused rollout kernels against the step kernel on the same configs."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv
from tests import replay
from tests.fuzz_configs import meta_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(18))
def test_hip_matches_oracle_under_a_fuzzed_config(seed):
    import torch

    import marl_sortingenv_amd as M

    kind = ("mono", "press", "sort")[seed % 3]
    masking = seed % 4 != 3 or kind == "sort"
    check_overflow = seed % 5 == 4
    hoard = seed % 6 == 5  # mostly no-ops: containers far beyond their capacity
    meta = meta_for(kind, seed)
    n, base = 64, 300 + seed
    cfg = replay.sorting_config(meta)
    env = M.BatchedSortingEnv(kind=kind, num_envs=n, device=0, base_seed=base, max_steps=meta["max_steps"],
                              noise_sorting=meta["noise_sorting"], balesize=meta["balesize"], config=cfg, auto_reset=True)
    oracles = [OracleEnv(kind=kind, seed=base + i, cfg=replay.oracle_config(meta)) for i in range(n)]
    obs0 = env.obs.cpu().numpy()
    for i, o in enumerate(oracles):
        assert np.array_equal(obs0[i].view(np.uint32), o.reset(base + i).view(np.uint32)), i
    g = torch.Generator(device="cpu").manual_seed(seed)
    for t in range(int(2.5 * meta["max_steps"])):
        mask = env.action_masks().cpu().numpy()
        if masking:
            act = torch.multinomial(torch.as_tensor(mask, dtype=torch.float32), 1, generator=g).squeeze(1).to(torch.int32)
        else:
            act = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        if hoard:
            keep = torch.rand(n, generator=g) < 0.1
            noop = (act // 11) * 11 if kind == "mono" else (torch.zeros_like(act) if kind == "press" else act)
            act = torch.where(keep, act, noop)
        obs, rew, done, m2 = env.step(act, use_action_masking=masking, check_overflow=check_overflow, want_reward64=True)
        obs, done, m2, r64 = obs.cpu().numpy(), done.cpu().numpy(), m2.cpu().numpy(), env.reward64.cpu().numpy()
        for i, o in enumerate(oracles):
            assert np.array_equal(mask[i], o.action_masks()), (t, i)
            eo, er, et = o.step(int(act[i]), use_action_masking=masking, check_overflow=check_overflow)
            if et:
                eo = o.reset(None)
            assert abs(er - r64[i]) <= 1e-6 and bool(done[i]) == et, (t, i, er, r64[i])
            assert np.array_equal(obs[i].view(np.uint32), eo.view(np.uint32)), (t, i, obs[i], eo)
            assert np.array_equal(m2[i], o.action_masks()), (t, i)
    ints, dbls, rng = (x.cpu().numpy() for x in env.get_state())
    for i, o in enumerate(oracles):
        I, D, R = o.snapshot()
        assert np.array_equal(ints[i, :56], I[:56]) and np.array_equal(dbls[i], D[:4]), i
        assert np.array_equal(rng[i].view(np.uint64)[0:4], R[0:4]), i
    assert env.error_count() == 0


@pytest.mark.parametrize("seed", range(6))
def test_rollout_kernels_match_the_step_kernel_under_a_fuzzed_config(seed):
    """whatever rollout kernel the handle picks for the config (three-role ring, two-role, one-lane, general generator)
    equals single steps fed with its actions"""
    import torch

    import marl_sortingenv_amd as M

    kind = ("mono", "press", "sort")[seed % 3]
    meta = meta_for(kind, seed + 20)
    kw = dict(kind=kind, num_envs=900, device=0, base_seed=7, max_steps=meta["max_steps"], noise_sorting=meta["noise_sorting"],
              balesize=meta["balesize"], config=replay.sorting_config(meta), auto_reset=True)
    a, b = M.BatchedSortingEnv(**kw), M.BatchedSortingEnv(**kw)
    K = 45
    buf = a.rollout(K, policy_seed=3)
    for k in range(K):
        obs, rew, done, mask = b.step(buf["actions"][k])
        assert torch.equal(obs, buf["obs"][k]) and torch.equal(rew, buf["reward"][k]), k
        assert torch.equal(done, buf["done"][k]) and torch.equal(mask, buf["mask"][k]), k
    for x, y in zip(a.get_state(), b.get_state()):
        assert torch.equal(x, y)


@pytest.mark.parametrize("seed", range(4))
def test_model_mode_draws_match_oracle_under_a_fuzzed_config(seed):
    """mse_model_actions (Env_3 mode='model' without agents) vs the oracle's restatement, 64 envs, fuzzed configs"""
    import torch

    import marl_sortingenv_amd as M

    masking = seed % 2 == 0
    meta = meta_for("mono", 40 + seed)
    n, base = 64, 900 + seed
    env = M.BatchedSortingEnv(kind="mono", num_envs=n, device=0, base_seed=base, max_steps=meta["max_steps"],
                              noise_sorting=meta["noise_sorting"], balesize=meta["balesize"],
                              config=replay.sorting_config(meta), auto_reset=True)
    oracles = [OracleEnv(kind="mono", seed=base + i, cfg=replay.oracle_config(meta)) for i in range(n)]
    for i, o in enumerate(oracles):
        o.reset(base + i)
    for t in range(2 * meta["max_steps"]):
        act = env.model_actions(masking)
        obs, rew, done, _ = env.step(act, want_reward64=True)
        a_h, obs, r64 = act.cpu().numpy(), obs.cpu().numpy(), env.reward64.cpu().numpy()
        for i, o in enumerate(oracles):
            assert o.model_fallback_action(masking) == int(a_h[i]), (t, i)
            eo, er, et = o.step(int(a_h[i]))
            if et:
                eo = o.reset(None)
            assert abs(er - r64[i]) <= 1e-6 and np.array_equal(obs[i].view(np.uint32), eo.view(np.uint32)), (t, i)
    rng = env.get_state()[2].cpu().numpy().view(np.uint64)
    for i, o in enumerate(oracles):
        assert np.array_equal(rng[i, 12:24], o.snapshot()[2][12:24]), i  # rng_pressing and rng_sorting, buffers included
