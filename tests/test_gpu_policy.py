"""PARITY (GPU): mse_policy_forward against a plain PyTorch fp32 restatement of the reference's policy
(src/training.py:115-131: MaskableActorCriticPolicy, net_arch=dict(pi=[32,32], vf=[32,32]), tanh; sb3_contrib's
MaskableCategorical: masked logits = -1e8, Categorical(logits=...)).

Floating point, so tolerances (written here): logits / value 2e-5 absolute (MFMA sums in a permuted order, tanh by
v_exp_f32 + v_rcp_f32), log-probabilities 1e-4.  stable_baselines3 is not installed in this image and the
reference ships no policy fixture, so the network itself is "parity unpinned" beyond this torch reference of the
same op; the weights are random with SB3's state_dict names and shapes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGIT_TOL, LOGP_TOL = 2e-5, 1e-4


def _weights(obs_dim, n_actions, seed):
    import torch

    from marl_sortingenv_amd.policy import SB3_KEYS, _shapes

    g = torch.Generator().manual_seed(seed)
    return {k: (torch.randn(s, generator=g) * (0.5 if len(s) == 2 else 0.1)).float()
            for k, s in zip(SB3_KEYS, _shapes(obs_dim, n_actions))}


def _torch_reference(w, obs, mask):
    """obs [N, D] f32 cpu, mask [N, A] bool cpu or None -> masked logits, log-softmax, value (all fp32, CPU)."""
    import torch
    import torch.nn.functional as F

    hp = torch.tanh(F.linear(torch.tanh(F.linear(obs, w["mlp_extractor.policy_net.0.weight"],
                                                 w["mlp_extractor.policy_net.0.bias"])),
                             w["mlp_extractor.policy_net.2.weight"], w["mlp_extractor.policy_net.2.bias"]))
    hv = torch.tanh(F.linear(torch.tanh(F.linear(obs, w["mlp_extractor.value_net.0.weight"],
                                                 w["mlp_extractor.value_net.0.bias"])),
                             w["mlp_extractor.value_net.2.weight"], w["mlp_extractor.value_net.2.bias"]))
    logits = F.linear(hp, w["action_net.weight"], w["action_net.bias"])
    if mask is not None:
        logits = torch.where(mask, logits, torch.tensor(-1e8))
    value = F.linear(hv, w["value_net.weight"], w["value_net.bias"]).squeeze(1)
    return logits, torch.log_softmax(logits, dim=1), value


def _fmix32(h):
    M = 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M
    h ^= h >> 16
    return h


def _word(seed, g, t):
    """mse_policy_word(mse_policy_key(seed, g), t) of csrc/mse_policy_stream.h on the host."""
    M = 0xFFFFFFFF
    s = _fmix32((seed & M) ^ _fmix32(((seed >> 32) + 0x9E3779B9) & M))
    key = _fmix32((s + (g & M) * 0x9E3779B1 + (g >> 32) * 0xC2B2AE3D) & M)
    c = ((t & M) * 0x85EBCA77 + (t >> 32) * 0x27D4EB2F) & M
    return _fmix32(c ^ key)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("obs_dim,n_actions,n", [(29, 22, 1000), (16, 11, 333), (13, 2, 64), (29, 22, 31)])
def test_policy_forward_matches_torch_fp32(obs_dim, n_actions, n, precision):
    import torch

    import marl_sortingenv_amd as M

    w = _weights(obs_dim, n_actions, seed=obs_dim * 100 + n_actions)
    pol = M.MlpPolicy(obs_dim, n_actions, w, device=0, precision=precision)
    assert pol.precision == precision
    g = torch.Generator().manual_seed(5)
    obs = torch.rand((n, obs_dim), generator=g)  # observations live in [0, 1] (clip in env_super.py:339-359)
    mask = torch.rand((n, n_actions), generator=g) < 0.6
    mask[:, 0] = True  # action 0 is always valid in the reference's masks
    for use_mask in (True, False):
        mk = mask if use_mask else None
        ref_logits, ref_logsm, ref_value = _torch_reference(w, obs, mk)
        out = pol.forward(obs.cuda(), None if mk is None else mk.cuda(), seed=77, t=3, want_logits=True)
        logits = out["logits"].cpu()
        assert torch.allclose(logits, ref_logits, atol=LOGIT_TOL, rtol=1e-6), (logits - ref_logits).abs().max()
        assert torch.allclose(out["value"].cpu(), ref_value, atol=LOGIT_TOL, rtol=1e-6)
        act = out["action"].cpu().long()
        assert bool(((act >= 0) & (act < n_actions)).all())
        if mk is not None:
            assert bool(mk.gather(1, act.unsqueeze(1)).all()), "a masked action was sampled"
        # log-probability of the sampled action
        assert torch.allclose(out["logp"].cpu(), ref_logsm.gather(1, act.unsqueeze(1)).squeeze(1), atol=LOGP_TOL)
        # the sample is the inverse cdf of the engine's stream over the softmax masses in REGISTER order (the MFMA
        # accumulator's rows of half 0, then half 1: csrc/mse_policy_device.h): recompute it in fp64 from the device's
        # own logits; a draw that lands within 1e-5 of a boundary may fall on either side
        order = [a for h in (0, 1) for r in range(16) for a in [(r & 3) + 8 * (r >> 2) + 4 * h] if a < n_actions]
        assert sorted(order) == list(range(n_actions))
        p = torch.softmax(logits.double(), dim=1).numpy()[:, order]
        cdf = np.cumsum(p, axis=1)
        for i in range(n):
            u = (_word(77, i, 3) >> 8) * 2.0 ** -24
            k = int(np.searchsorted(cdf[i], u, side="right"))
            k = order[min(k, n_actions - 1)]
            if int(act[i]) != k:
                near = min(abs(u - cdf[i, j]) for j in range(n_actions))
                assert near < 1e-5, (i, int(act[i]), k, u, cdf[i])
        # deterministic = argmax of the masked logits
        det = pol.forward(obs.cuda(), None if mk is None else mk.cuda(), deterministic=True)["action"].cpu().long()
        top2 = torch.topk(ref_logits, 2, dim=1).values if n_actions > 1 else None
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert bool((det[clear] == ref_logits.argmax(dim=1)[clear]).all())


def test_policy_sampling_frequencies_follow_the_softmax():
    """One observation, 20 000 steps of the stream: empirical action frequencies vs softmax probabilities."""
    import torch

    import marl_sortingenv_amd as M

    obs_dim, n_actions, n = 29, 22, 20000
    w = _weights(obs_dim, n_actions, seed=9)
    pol = M.MlpPolicy(obs_dim, n_actions, w, device=0)
    obs = torch.rand((1, obs_dim), generator=torch.Generator().manual_seed(1)).repeat(n, 1)
    mask = torch.ones((n, n_actions), dtype=torch.bool)
    mask[:, [3, 7, 20]] = False
    out = pol.forward(obs.cuda(), mask.cuda(), seed=123, t=0)  # env index varies: n independent draws
    freq = torch.bincount(out["action"].cpu().long(), minlength=n_actions).double() / n
    _, logsm, _ = _torch_reference(w, obs[:1], mask[:1])
    p = logsm.exp().double().squeeze(0)
    assert float(freq[[3, 7, 20]].sum()) == 0.0
    assert float((freq - p).abs().max()) < 4.0 * float((p * (1 - p) / n).sqrt().max()) + 1e-3


def test_policy_drives_the_step_engine():
    """policy.forward -> env.step loop stays on the device and only ever takes valid actions."""
    import torch

    import marl_sortingenv_amd as M

    n = 4096
    env = M.BatchedSortingEnv(kind="mono", num_envs=n, device=0, base_seed=3, max_steps=50, noise_sorting=0.0)
    pol = M.MlpPolicy(env.obs_dim, env.num_actions, _weights(env.obs_dim, env.num_actions, seed=4), device=0)
    obs, mask = env.obs, env.mask
    total = torch.zeros((n,), dtype=torch.float64, device="cuda")
    for t in range(60):
        out = pol.forward(obs, mask, seed=11, t=t)
        assert bool(torch.gather(mask, 1, out["action"].long().unsqueeze(1)).all())
        obs, rew, done, mask = env.step(out["action"])
        total += rew.double()
    assert env.error_count() == 0 and bool(torch.isfinite(total).all())


def test_policy_abi_errors():
    import ctypes as C

    import marl_sortingenv_amd as M

    L = M.load_library()
    h = C.c_void_p()
    buf = (C.c_float * 8)()
    assert L.mse_policy_create(C.byref(h), 64, 22, buf, 0) == -2  # MSE_ERR_UNSUPPORTED_CONFIG
    assert L.mse_policy_create(C.byref(h), 29, 22, None, 0) == -1
    assert b"mse_policy_create" in L.mse_last_error()
    assert L.mse_policy_num_weights(29, 22) == 2 * (32 * 29 + 32 + 32 * 32 + 32) + 22 * 32 + 22 + 32 + 1


@pytest.mark.parametrize("kind", ["mono", "press"])
def test_collector_fills_rollout_buffers_consistently(kind):
    """PolicyRolloutCollector: the buffers replay - stepping a twin env with the recorded actions (and, for Env_2,
    the sorting decisions of the same sorting policy) reproduces the recorded observations, masks and rewards; the
    recorded log-probabilities / values are those of the recorded observations."""
    import torch

    import marl_sortingenv_amd as M

    n, K = 2000, 24
    kw = dict(kind=kind, num_envs=n, device=0, base_seed=17, max_steps=10, noise_sorting=0.05, balesize=200)
    env, twin = M.BatchedSortingEnv(**kw), M.BatchedSortingEnv(**kw)
    pol = M.MlpPolicy(env.obs_dim, env.num_actions, _weights(env.obs_dim, env.num_actions, seed=21), device=0)
    sort_pol = M.MlpPolicy(13, 2, _weights(13, 2, seed=22), device=0) if kind == "press" else None
    col = M.PolicyRolloutCollector(env, pol, K, sort_policy=sort_pol, seed=5)
    out = col.collect()
    assert int(out["episode_starts"][0].sum()) == n and int(out["episode_starts"].sum()) == n * (1 + (K - 1) // 10)
    for k in range(K):
        assert torch.equal(out["observations"][k], twin.obs) and torch.equal(out["action_masks"][k], twin.mask)
        again = pol.forward(twin.obs, twin.mask, seed=5, t=k)
        assert torch.equal(again["action"], out["actions"][k])
        assert torch.equal(again["logp"], out["log_probs"][k]) and torch.equal(again["value"], out["values"][k])
        assert bool(torch.gather(twin.mask, 1, out["actions"][k].long().unsqueeze(1)).all())
        sm = sort_pol.forward(twin.sort_agent_obs(), None, deterministic=True)["action"] if sort_pol else None
        _, rew, done, _ = twin.step(out["actions"][k], sort_mode=sm)
        assert torch.equal(rew, out["rewards"][k])
    assert torch.equal(out["last_dones"], twin.done)
    assert env.error_count() == 0


def test_f16x3_form_tracks_the_exact_f32_form():
    """The split form (three f16 MFMAs per product on 22-bit operand pairs) against the exact f32 form of the same
    kernel: logits (magnitudes up to ~6 here) and values within 1.2e-5 - measured 6.7e-6, of which ~2e-6 is the f32
    form's own summation-order noise - inside the 2e-5 the parity test allows either form against PyTorch; a policy
    whose folded weights exceed f16's range falls back to f32 by itself."""
    import torch

    import marl_sortingenv_amd as M

    w = _weights(29, 22, seed=77)
    obs = torch.rand((4096, 29), generator=torch.Generator().manual_seed(3)).cuda()
    exact = M.MlpPolicy(29, 22, w, device=0, precision="f32").forward(obs, None, want_logits=True, deterministic=True)
    split = M.MlpPolicy(29, 22, w, device=0, precision="f16x3").forward(obs, None, want_logits=True, deterministic=True)
    assert float((exact["logits"] - split["logits"]).abs().max()) < 1.2e-5
    assert float((exact["value"] - split["value"]).abs().max()) < 1.2e-5
    big = dict(w)
    big["mlp_extractor.policy_net.2.weight"] = w["mlp_extractor.policy_net.2.weight"] * 3.0e4
    pol = M.MlpPolicy(29, 22, big, device=0)
    assert pol.precision == "f32"
    with pytest.raises(M.MseError):
        pol.set_precision("f16x3")
    assert M.MlpPolicy(29, 22, w, device=0).precision == "f16x3"  # auto


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("kind,noise,n,K", [("mono", 0.0, 5000, 24), ("mono", 0.05, 777, 23), ("press", 0.05, 1030, 21),
                                            ("sort", 0.0, 600, 25), ("mono", 0.0, 70300, 6), ("mono", 0.05, 1, 14),
                                            ("press", 0.0, 33, 14),
                                            # whole workgroups, and the 65 536 envs of the headline batch
                                            ("mono", 0.0, 1024, 24), ("mono", 0.05, 512, 23), ("press", 0.05, 256, 21),
                                            ("sort", 0.0, 768, 25), ("mono", 0.0, 65536, 7)])
def test_fused_policy_rollout_equals_two_launch_collector(kind, noise, n, K, precision):
    """mse_rollout_policy (policy forward inside the rollout kernel, one launch per rollout) fills the same
    MaskableRolloutBuffer-shaped tensors, bit for bit, as alternating mse_policy_forward and mse_step; the env state
    afterwards is the same too, over several consecutive rollouts (ragged N, auto-resets inside a rollout)."""
    import torch

    import marl_sortingenv_amd as M

    kw = dict(kind=kind, num_envs=n, device=0, base_seed=17, max_steps=10, noise_sorting=noise, balesize=200)
    a, b = M.BatchedSortingEnv(**kw), M.BatchedSortingEnv(**kw)
    pol = M.MlpPolicy(a.obs_dim, a.num_actions, _weights(a.obs_dim, a.num_actions, seed=31), device=0, precision=precision)
    sm = (torch.arange(n) % 2).to(torch.int32).cuda() if kind == "press" else None
    two = M.PolicyRolloutCollector(a, pol, K, seed=5)
    if sm is not None:  # the two-launch collector takes Env_2's sorting decisions from a policy; feed it the tensor
        class Frozen:
            def forward(self, *args, **kwargs):
                return {"action": sm}
        two.sort_policy = Frozen()
    fused = M.FusedPolicyRollout(b, pol, K, seed=5, sort_mode=sm)
    for it in range(3):
        x, y = two.collect(), fused.collect()
        for key in ("observations", "action_masks", "episode_starts", "actions", "log_probs", "values", "rewards",
                    "last_values", "last_dones"):
            assert torch.equal(x[key], y[key]), (it, key)
        for sa, sb in zip(a.get_state(), b.get_state()):
            assert torch.equal(sa, sb), it
    assert a.policy_step == b.policy_step == 3 * K and b.error_count() == 0
    # deterministic (argmax) collection agrees as well
    x, y = two.collect(deterministic=True), fused.collect(deterministic=True)
    assert torch.equal(x["actions"], y["actions"]) and torch.equal(x["rewards"], y["rewards"])


@pytest.mark.parametrize("pipeline", [1, 2])
@pytest.mark.parametrize("kind,noise,n,K,flags", [("mono", 0.0, 5000, 24, 0), ("mono", 0.05, 777, 23, 0), ("press", 0.05, 1030, 21, 0),
                                                  ("sort", 0.05, 600, 25, 0), ("mono", 0.05, 1, 14, 0), ("press", 0.0, 33, 14, 0),
                                                  ("mono", 0.0, 257, 30, 1), ("mono", 0.0, 1500, 40, 3), ("mono", 0.0, 65536, 7, 0)])
def test_fused_policy_rollout_kernel_shapes_agree(kind, noise, n, K, flags, pipeline):
    """Up to 256 envs x CUs the learned-policy rollout has three kernel shapes: actor + critic + RNG waves per 64 envs
    (by size, rollout_pipeline 0: the one the test above holds to the two-launch collector), actor + critic waves (1)
    and one wave per 32 envs doing everything (2).  All three must collect the same bits and leave the same state -
    also unmasked (flags 1) and with check_overflow (3), over auto-resets and with a one-step and a bootstrap-only tail."""
    import torch

    import marl_sortingenv_amd as M

    kw = dict(kind=kind, num_envs=n, device=0, base_seed=29, max_steps=12, noise_sorting=noise, balesize=200)
    a, b = M.BatchedSortingEnv(rollout_pipeline=0, **kw), M.BatchedSortingEnv(rollout_pipeline=pipeline, **kw)
    pol = M.MlpPolicy(a.obs_dim, a.num_actions, _weights(a.obs_dim, a.num_actions, seed=37), device=0, precision="f16x3")
    sm = (torch.arange(n) % 2).to(torch.int32).cuda() if kind == "press" else None
    fa = M.FusedPolicyRollout(a, pol, K, seed=6, sort_mode=sm)
    fb = M.FusedPolicyRollout(b, pol, K, seed=6, sort_mode=sm)
    how = dict(use_action_masking=not (flags & 1), check_overflow=bool(flags & 2))
    for it, steps in enumerate((K, K, 1, K)):
        x, y = fa.collect(steps, **how), fb.collect(steps, **how)
        for key in x:
            assert torch.equal(x[key], y[key]), (it, key)  # (rows past `steps` keep the previous, equal, rollout)
        for sa, sb in zip(a.get_state(), b.get_state()):
            assert torch.equal(sa, sb), it
    x, y = fa.collect(deterministic=True, **how), fb.collect(deterministic=True, **how)
    for key in x:
        assert torch.equal(x[key], y[key]), ("deterministic", key)
    assert a.error_count() == 0 and b.error_count() == 0


@pytest.mark.parametrize("kind", ["mono", "sort"])
def test_fused_policy_rollout_kernel_shapes_agree_without_bale_tracking(kind):
    """track_bales=False: the roles kernel posts no bale bookings (and, for Env_1, nothing at all besides the rows)."""
    import torch

    import marl_sortingenv_amd as M

    kw = dict(kind=kind, num_envs=700, device=0, base_seed=41, max_steps=9, noise_sorting=0.05, balesize=200, track_bales=False)
    a, b = M.BatchedSortingEnv(rollout_pipeline=0, **kw), M.BatchedSortingEnv(rollout_pipeline=2, **kw)
    pol = M.MlpPolicy(a.obs_dim, a.num_actions, _weights(a.obs_dim, a.num_actions, seed=38), device=0, precision="f16x3")
    fa, fb = M.FusedPolicyRollout(a, pol, 20, seed=6), M.FusedPolicyRollout(b, pol, 20, seed=6)
    for it in range(2):
        x, y = fa.collect(), fb.collect()
        for key in x:
            assert torch.equal(x[key], y[key]), (it, key)
        for sa, sb in zip(a.get_state(), b.get_state()):
            assert torch.equal(sa, sb), it


@pytest.mark.parametrize("n,noise", [(3000, 0.05), (70300, 0.0)])
def test_fused_rollout_with_the_sorting_agent_in_the_loop(n, noise):
    """Env_2_Pressing with its pre-trained sorting agent (env_2_press.py:101-104) as a second network inside the
    rollout kernel == the two-launch collector, which evaluates that network on mse_sort_agent_obs every step."""
    import torch

    import marl_sortingenv_amd as M

    K = 12
    kw = dict(kind="press", num_envs=n, device=0, base_seed=23, max_steps=9, noise_sorting=noise, balesize=200)
    a, b = M.BatchedSortingEnv(**kw), M.BatchedSortingEnv(**kw)
    pol = M.MlpPolicy(16, 11, _weights(16, 11, seed=41), device=0)
    sort_pol = M.MlpPolicy(13, 2, _weights(13, 2, seed=42), device=0)
    two = M.PolicyRolloutCollector(a, pol, K, sort_policy=sort_pol, seed=8)
    fused = M.FusedPolicyRollout(b, pol, K, seed=8, sort_policy=sort_pol)
    for it in range(3):
        x, y = two.collect(), fused.collect()
        for key in ("observations", "action_masks", "episode_starts", "actions", "log_probs", "values", "rewards",
                    "last_values", "last_dones"):
            assert torch.equal(x[key], y[key]), (it, key)
        for sa, sb in zip(a.get_state(), b.get_state()):
            assert torch.equal(sa, sb), it
    # the sorting decisions matter: a different sorting network gives different rollouts
    other = M.FusedPolicyRollout(M.BatchedSortingEnv(**kw), pol, K, seed=8, sort_policy=M.MlpPolicy(13, 2, _weights(13, 2, seed=43), device=0))
    fresh = M.FusedPolicyRollout(M.BatchedSortingEnv(**kw), pol, K, seed=8, sort_policy=sort_pol)
    assert not torch.equal(other.collect()["rewards"], fresh.collect()["rewards"])
