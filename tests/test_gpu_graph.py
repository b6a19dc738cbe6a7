"""GPU: the C ABI's launches are graph-capture safe (include/mse.h: mse_step / mse_rollout / mse_reset enqueue on the
caller's stream, allocate nothing and never synchronise).  A torch.cuda.CUDAGraph (= hipGraph) captures K mse_step
launches fed from a static action buffer; replaying it must equal eager stepping.  (Launches that draw from the policy
stream take their step counter from the host handle at enqueue time, so a captured mse_rollout replays the same draws:
capture those per rollout, or step with explicit actions as here.)"""
import pytest

pytestmark = pytest.mark.gpu


def test_captured_steps_replay_like_eager_steps():
    import torch

    import marl_sortingenv_amd as M

    n, K, rounds = 4096, 8, 5
    kw = dict(kind="mono", num_envs=n, device=0, base_seed=21, max_steps=30, noise_sorting=0.05, balesize=200)
    eager, graphed = M.BatchedSortingEnv(**kw), M.BatchedSortingEnv(**kw)
    g = torch.Generator(device="cpu").manual_seed(1)
    static_actions = torch.zeros((K, n), dtype=torch.int32, device="cuda")
    out_obs = torch.zeros((K, n, 29), dtype=torch.float32, device="cuda")
    out_rew = torch.zeros((K, n), dtype=torch.float32, device="cuda")

    def body(env):
        for k in range(K):  # actions ignore the mask on purpose: the step accepts any action in range
            obs, rew, done, mask = env.step(static_actions[k])
            out_obs[k].copy_(obs)
            out_rew[k].copy_(rew)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on the capture stream (torch's recipe), on a throw-away handle
        warm = M.BatchedSortingEnv(**kw)
        body(warm)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body(graphed)
    # the capture itself does not execute: the graphed env is still in its reset state
    for a, b in zip(graphed.get_state(), eager.get_state()):
        assert torch.equal(a, b)
    for r in range(rounds):
        acts = torch.randint(0, 22, (K, n), generator=g, dtype=torch.int32)
        static_actions.copy_(acts)
        graph.replay()
        torch.cuda.synchronize()
        got_obs, got_rew = out_obs.clone(), out_rew.clone()
        for k in range(K):
            obs, rew, done, mask = eager.step(acts[k].cuda())
            assert torch.equal(obs, got_obs[k]) and torch.equal(rew, got_rew[k]), (r, k)
    for a, b in zip(graphed.get_state(), eager.get_state()):
        assert torch.equal(a, b)
    assert graphed.error_count() == 0
