"""Generate tests/golden/*.npz from the imported reference (build container only).

    python -m oracle.gen_golden            # from the repo root

Each fixture is DATA: the scripted inputs (ops, actions, sort modes, flags) and what the
reference returned / held after every op (obs f32, reward f64, terminated, action mask, integer
state, accuracy doubles, PCG64 states of the three live streams).  Nothing of the reference's
source is stored.  Layout of the integer snapshot: oracle/oracle.py SNAP.

Row ops:  op=0  step(action[, sort_mode], flags)      op=1  reset(seed=arg)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from oracle.oracle import STEP_CHECK_OVERFLOW, STEP_UNMASKED  # noqa: E402

OUT_DIR = os.path.join(ROOT, "tests", "golden")

POLICIES = ("masked_uniform", "noop", "wrong_mode", "ignore_mask_uniform", "greedy_press")


def scenarios():
    """(name, kind, ctor kwargs, seeds for the reset boundaries, policy, masking, check_overflow, steps/segment)"""
    S = []
    big = (1 << 32) - 3  # seed+3/+4/+99 need two SeedSequence words; the reference caps seeds at 2**32-1 (np.random.seed)
    for kind in ("mono", "press", "sort"):
        for noise in (0.0, 0.05):
            ntag = "n0" if noise == 0.0 else "n5"
            S.append((f"{kind}_{ntag}_masked_s0_s3", kind, dict(max_steps=200, noise_sorting=noise, balesize=200),
                      [0, 3], "masked_uniform", True, False, 215))
            S.append((f"{kind}_{ntag}_masked_s1_s42", kind, dict(max_steps=200, noise_sorting=noise, balesize=200),
                      [1, 42], "masked_uniform", True, False, 215))
            S.append((f"{kind}_{ntag}_ignoremask_on_s42_s7", kind, dict(max_steps=200, noise_sorting=noise, balesize=200),
                      [42, 7], "ignore_mask_uniform", True, False, 205))
            if kind != "sort":
                S.append((f"{kind}_{ntag}_unmasked_s3_s{big}", kind, dict(max_steps=200, noise_sorting=noise, balesize=200),
                          [3, big], "ignore_mask_uniform", False, False, 205))
        S.append((f"{kind}_n0_noop_s3", kind, dict(max_steps=200, noise_sorting=0.0, balesize=200),
                  [3], "noop", True, False, 200))
        S.append((f"{kind}_n0_wrongmode_s0", kind, dict(max_steps=200, noise_sorting=0.0, balesize=200),
                  [0, 123456789], "wrong_mode", True, False, 120))
        S.append((f"{kind}_n5_overflow_s1", kind, dict(max_steps=200, noise_sorting=0.05, balesize=200),
                  [1, 2], "noop", True, True, 90))
        S.append((f"{kind}_n5_greedy_s5_bale150_T50", kind, dict(max_steps=50, noise_sorting=0.05, balesize=150),
                  [5, 6, 99991], "greedy_press", True, False, 60))
    # a config.yml other than the reference's own (every value the step path reads that config.yml can change),
    # written to a scratch directory the reference env is constructed in
    for name, kind, kw, seeds, policy, masking, seg in (
            ("mono_cfgA_n5_masked_s2_s11", "mono", dict(max_steps=60, noise_sorting=0.05, balesize=150), [2, 11], "masked_uniform", True, 75),
            ("press_cfgA_n0_unmasked_s4_s8", "press", dict(max_steps=60, noise_sorting=0.0, balesize=150), [4, 8], "ignore_mask_uniform", False, 75),
            ("sort_cfgA_n5_masked_s6_s9", "sort", dict(max_steps=60, noise_sorting=0.05, balesize=150), [6, 9], "masked_uniform", True, 75),
            ("mono_cfgA_n0_greedy_s7", "mono", dict(max_steps=80, noise_sorting=0.0, balesize=150), [7, 13], "greedy_press", True, 85)):
        S.append((name, kind, kw, seeds, policy, masking, False, seg, CONFIG_A))
    # a batch size whose floor() leaves units over (0.35 * 90 = 31.49..., 0.15 * 90 = 13.5): the generator's private
    # stream becomes observable - remainder units by rng.choice, the shuffle's draws in between
    # (utils/input_generator.py:46-61); `rng` then holds five streams, the generator's last
    for name, kind, kw, seeds, policy, masking, seg in (
            ("mono_cfgB_n5_masked_s2_s11", "mono", dict(max_steps=60, noise_sorting=0.05, balesize=200), [2, 11], "masked_uniform", True, 75),
            ("press_cfgB_n0_unmasked_s4_s8", "press", dict(max_steps=60, noise_sorting=0.0, balesize=200), [4, 8], "ignore_mask_uniform", False, 75),
            ("sort_cfgB_n5_masked_s6_s9", "sort", dict(max_steps=60, noise_sorting=0.05, balesize=200), [6, 9], "masked_uniform", True, 75),
            ("mono_cfgC_n0_greedy_s7_batch77", "mono", dict(max_steps=80, noise_sorting=0.0, balesize=150), [7, 13], "greedy_press", True, 85)):
        S.append((name, kind, kw, seeds, policy, masking, False, seg, CONFIG_C if "cfgC" in name else CONFIG_B))
    return S


CONFIG_A = {
    "simulation": {"steps_per_pattern": 7},
    "sorting_station": {"baseline_accuracy": [0.7, 0.8, 0.65, 0.9], "boost": 0.2},
    "pressing_station": {"press_times": {1: 8, 2: 20}, "container_capacity": 500, "bale_remainder_threshold": 0.4,
                         "bale_quality_thresholds": {"A": 0.85, "B": 0.9, "C": 0.8, "D": 0.95}},
    "rewards": {"sorting": {"purity_threshold_theta": 0.75, "tanh_temperature": 0.7},
                "pressing": {"overflow_penalty_catastrophic": -0.9, "overflow_penalty_severe": -0.4,
                             "overflow_penalty_mild": -0.1, "bale_efficiency_factor": 0.8, "max_state_reward": 0.4},
                "overflow_termination_penalty": -7.5},
}


CONFIG_B = {"simulation": {"input_batch_size": 90}}
CONFIG_C = {"simulation": {"input_batch_size": 77}, "sorting_station": {"baseline_accuracy": [0.7, 0.8, 0.65, 0.9], "boost": 0.2}}


def _construct(cls, seed, kw, overrides):
    """Env(...) with the reference's config.yml, or with a merged copy of it in a scratch CWD."""
    if not overrides:
        return cls(seed=seed, **kw)
    import tempfile

    import yaml

    def merge(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                merge(a[k], v)
            else:
                a[k] = v
        return a

    with open(os.path.join(ref_harness.REFERENCE_ROOT, "config.yml")) as f:
        cfg = merge(yaml.safe_load(f), overrides)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "config.yml"), "w") as f:
            yaml.safe_dump(cfg, f)
        os.chdir(d)
        try:
            return cls(seed=seed, **kw)  # reads ./config.yml (env_super.py:25-29)
        finally:
            os.chdir(cwd)


def choose(policy, kind, env, prng, t):
    """Returns (action, sort_mode) for the next step; sort_mode only matters for 'press'."""
    mask = np.asarray(env.action_masks(), dtype=bool)
    n_act = mask.size
    sort_mode = int(prng.integers(0, 2))
    if policy == "masked_uniform":
        a = int(prng.choice(np.flatnonzero(mask)))
    elif policy == "ignore_mask_uniform":
        a = int(prng.integers(0, n_act))
    elif policy == "noop":
        a = 0
        sort_mode = int((t // 7) % 2)
        if kind == "sort":
            a = sort_mode
        elif kind == "mono":
            a = 11 * sort_mode
    elif policy == "wrong_mode":
        # boost the group that is NOT dominant on the belt
        belt = env.current_material_belt
        wrong = 1 if (belt[0] + belt[2]) > (belt[1] + belt[3]) else 0
        sort_mode = wrong
        if kind == "sort":
            a = wrong
        else:
            valid = np.flatnonzero(mask[:11]) if kind == "mono" else np.flatnonzero(mask)
            p = int(prng.choice(valid))
            a = p + 11 * wrong if kind == "mono" else p
    elif policy == "greedy_press":
        # press whenever the mask allows it (highest valid press action), alternate modes
        sort_mode = int(t % 2)
        if kind == "sort":
            a = sort_mode
        else:
            valid = np.flatnonzero(mask[:11]) if kind == "mono" else np.flatnonzero(mask)
            p = int(valid[-1])
            a = p + 11 * sort_mode if kind == "mono" else p
    else:
        raise ValueError(policy)
    return a, sort_mode


def run_scenario(sc):
    name, kind, kw, seeds, policy, masking, check_overflow, seg = sc[:8]
    overrides = sc[8] if len(sc) > 8 else None
    classes = ref_harness.load()
    env = _construct(classes[kind], seeds[0], kw, overrides)
    agent = None
    if kind == "press":
        agent = ref_harness.ScriptedSortAgent()
        env.set_agents(sort_agent=agent)
    prng = np.random.default_rng(sum(name.encode()) * 7919)  # policy stream, independent of the env
    D = env.observation_space.shape[0]
    A = env.action_space.n
    rows = {k: [] for k in ("op", "arg", "sort_mode", "flags", "obs", "reward", "terminated", "mask",
                            "ints", "dbls", "rng")}
    agent_obs = []  # press only: the observation Env_2.step gave its sorting agent (env_2_press.py:101-104)

    def record(op, arg, sm, flags, obs, rew, term):
        I, Dd, R = ref_harness.snapshot(env)
        rows["op"].append(op)
        rows["arg"].append(arg)
        rows["sort_mode"].append(sm)
        rows["flags"].append(flags)
        rows["obs"].append(np.asarray(obs, dtype=np.float32))
        rows["reward"].append(float(rew))
        rows["terminated"].append(int(term))
        rows["mask"].append(np.asarray(env.action_masks(), dtype=np.uint8))
        rows["ints"].append(I)
        rows["dbls"].append(Dd)
        rows["rng"].append(R)

    flags = (0 if masking else STEP_UNMASKED) | (STEP_CHECK_OVERFLOW if check_overflow else 0)
    for seed in seeds:
        obs, _ = env.reset(seed=seed)
        record(1, seed, -1, 0, obs, 0.0, 0)
        agent_obs.append(np.zeros(13, dtype=np.float32))
        for t in range(seg):
            a, sm = choose(policy, kind, env, prng, t)
            kwargs = dict(use_action_masking=masking, check_overflow=check_overflow)
            if kind == "press":
                agent.next_mode = sm
            obs, rew, term, trunc, info = env.step(a, **kwargs)
            assert trunc is False
            record(0, a, sm if kind == "press" else -1, flags, obs, rew, term)
            agent_obs.append(agent.last_obs if kind == "press" else np.zeros(13, dtype=np.float32))
            if term and check_overflow:
                break  # overflow-terminated: go to the next seeded reset
    out = {k: np.asarray(v) for k, v in rows.items()}
    out["op"] = out["op"].astype(np.uint8)
    out["arg"] = out["arg"].astype(np.uint64)
    out["sort_mode"] = out["sort_mode"].astype(np.int8)
    out["flags"] = out["flags"].astype(np.uint8)
    out["terminated"] = out["terminated"].astype(np.uint8)
    out["rng"] = out["rng"].astype(np.uint64)
    assert out["obs"].shape[1] == D and out["mask"].shape[1] == A
    if kind == "press":
        out["agent_obs"] = np.asarray(agent_obs, dtype=np.float32)
        assert out["agent_obs"].shape == (len(out["op"]), 13)
    meta = dict(name=name, kind=kind, max_steps=kw["max_steps"], noise_sorting=kw["noise_sorting"],
                balesize=kw["balesize"], ctor_seed=int(seeds[0]), policy=policy, masking=bool(masking),
                check_overflow=bool(check_overflow), numpy=np.__version__, config_overrides=overrides or {},
                generator="oracle/gen_golden.py from the imported reference")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return name, out


def main():
    if not ref_harness.available():
        raise SystemExit("reference checkout not available: fixtures can only be generated in the build container")
    os.makedirs(OUT_DIR, exist_ok=True)
    total = 0
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for sc in scenarios():
        if only and not any(sc[0].startswith(o) for o in only):
            continue
        name, out = run_scenario(sc)
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        sz = os.path.getsize(path)
        total += sz
        print(f"{name:48s} rows={len(out['op']):4d}  {sz/1024:7.1f} KiB")
    print(f"total {total/1024:.1f} KiB")


if __name__ == "__main__" and not any(a in sys.argv for a in ("--rule-based", "--model", "--trace", "--random-mode", "--agents",
                                                              "--overflow-info")):
    main()


def gen_rule_based(seeds=range(1, 11), steps=200):
    """Env_3_Monolith.step(mode='rule_based') (env_monolith.py:166-184): the reference's own rule policy
    (sorting_rules + check_container_level) for the paper's benchmark seeds; actions and rewards per step."""
    cls = ref_harness.load()["mono"]
    acts, rews = [], []
    for s in seeds:
        env = cls(max_steps=steps, seed=s, noise_sorting=0.0, balesize=200)
        env.reset(seed=s)
        a_row, r_row = [], []
        for _ in range(steps):
            obs, r, term, trunc, info = env.step(action=None, mode="rule_based", use_action_masking=True)
            a_row.append(int(info["action"]))
            r_row.append(float(r))
        assert term
        acts.append(a_row)
        rews.append(r_row)
    out = dict(seeds=np.array(list(seeds), dtype=np.int64), actions=np.array(acts, dtype=np.int32),
               rewards=np.array(rews, dtype=np.float64))
    path = os.path.join(OUT_DIR, "rule_based_mono_n0_seeds1_10.npz")
    np.savez_compressed(path, **out)
    cum = out["rewards"].sum(1)
    print(f"rule_based: cumulative reward {cum.mean():.2f} +- {cum.std():.2f} over seeds {list(seeds)} -> {path}")


if __name__ == "__main__" and "--rule-based" in sys.argv:
    gen_rule_based()


def _pack(rows, meta):
    out = {k: np.asarray(v) for k, v in rows.items()}
    out["op"] = out["op"].astype(np.uint8)
    out["arg"] = out["arg"].astype(np.uint64)
    out["sort_mode"] = out["sort_mode"].astype(np.int8)
    out["flags"] = out["flags"].astype(np.uint8)
    out["terminated"] = out["terminated"].astype(np.uint8)
    out["rng"] = out["rng"].astype(np.uint64)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def gen_model_fallback():
    """Env_3_Monolith.step(action=None, mode='model') with NO agents assigned (env_monolith.py:186-221): the env
    draws rng_sorting.choice([0, 1]) and rng_pressing.choice(valid) / choice(11) itself.  Same row format as the
    step traces; `arg` of a step row is the action the reference reported in info["action"]; `rng` holds all four
    streams (rng, rng_noise, rng_pressing, rng_sorting)."""
    cls = ref_harness.load()["mono"]
    for name, masking, noise, seeds, seg in (("model_mono_n0_masked_s2_s17", True, 0.0, [2, 17], 210),
                                             ("model_mono_n5_unmasked_s4_s23", False, 0.05, [4, 23], 210)):
        env = cls(max_steps=200, seed=seeds[0], noise_sorting=noise, balesize=200)
        rows = {k: [] for k in ("op", "arg", "sort_mode", "flags", "obs", "reward", "terminated", "mask", "ints", "dbls", "rng")}

        def record(op, arg, obs, rew, term):
            I, Dd, R = ref_harness.snapshot(env)
            for k, v in (("op", op), ("arg", arg), ("sort_mode", -1), ("flags", 0), ("obs", np.asarray(obs, dtype=np.float32)),
                         ("reward", float(rew)), ("terminated", int(term)), ("mask", np.asarray(env.action_masks(), dtype=np.uint8)),
                         ("ints", I), ("dbls", Dd), ("rng", R)):
                rows[k].append(v)

        for seed in seeds:
            obs, _ = env.reset(seed=seed)
            record(1, seed, obs, 0.0, 0)
            for _ in range(seg):
                obs, rew, term, trunc, info = env.step(action=None, mode="model", use_action_masking=masking)
                record(0, int(info["action"]), obs, rew, term)
        meta = dict(name=name, kind="mono", max_steps=200, noise_sorting=noise, balesize=200, ctor_seed=int(seeds[0]),
                    policy="mode='model' without agents", masking=bool(masking), check_overflow=False,
                    numpy=np.__version__, config_overrides={}, generator="oracle/gen_golden.py --model from the imported reference")
        out = _pack(rows, meta)
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:48s} rows={len(out['op']):4d}  {os.path.getsize(path)/1024:7.1f} KiB")


def gen_trace():
    """One episode per fixture with the reference's per-env ledgers captured at its end: reward_data,
    press_actions_per_timestep and the full bale_count lists (what utils/plotting.py:32-48 reads)."""
    classes = ref_harness.load()
    for name, kind, kw, seed, policy, masking, check_overflow, steps in (
            ("trace_mono_n5_unmasked_s3", "mono", dict(max_steps=150, noise_sorting=0.05, balesize=200), 3, "ignore_mask_uniform", False, False, 150),
            ("trace_mono_n0_greedy_s5_bale150", "mono", dict(max_steps=120, noise_sorting=0.0, balesize=150), 5, "greedy_press", True, False, 120),
            ("trace_mono_n0_ignoremask_s8", "mono", dict(max_steps=150, noise_sorting=0.0, balesize=200), 8, "ignore_mask_uniform", True, False, 150),
            ("trace_press_n0_unmasked_s4", "press", dict(max_steps=150, noise_sorting=0.0, balesize=200), 4, "ignore_mask_uniform", False, False, 150),
            ("trace_sort_n5_masked_s6", "sort", dict(max_steps=150, noise_sorting=0.05, balesize=200), 6, "masked_uniform", True, False, 150),
            ("trace_mono_n5_overflow_s1", "mono", dict(max_steps=200, noise_sorting=0.05, balesize=200), 1, "noop", True, True, 200)):
        env = classes[kind](seed=seed, **kw)
        agent = None
        if kind == "press":
            agent = ref_harness.ScriptedSortAgent()
            env.set_agents(sort_agent=agent)
        prng = np.random.default_rng(sum(name.encode()) * 104729)
        rows = {k: [] for k in ("op", "arg", "sort_mode", "flags", "obs", "reward", "terminated", "mask", "ints", "dbls", "rng")}

        def record(op, arg, sm, flags, obs, rew, term):
            I, Dd, R = ref_harness.snapshot(env)
            for k, v in (("op", op), ("arg", arg), ("sort_mode", sm), ("flags", flags), ("obs", np.asarray(obs, dtype=np.float32)),
                         ("reward", float(rew)), ("terminated", int(term)), ("mask", np.asarray(env.action_masks(), dtype=np.uint8)),
                         ("ints", I), ("dbls", Dd), ("rng", R)):
                rows[k].append(v)

        flags = (0 if masking else STEP_UNMASKED) | (STEP_CHECK_OVERFLOW if check_overflow else 0)
        obs, _ = env.reset(seed=seed)
        record(1, seed, -1, 0, obs, 0.0, 0)
        for t in range(steps):
            a, sm = choose(policy, kind, env, prng, t)
            if kind == "press":
                agent.next_mode = sm
            obs, rew, term, trunc, info = env.step(a, use_action_masking=masking, check_overflow=check_overflow)
            record(0, a, sm if kind == "press" else -1, flags, obs, rew, term)
            if term:
                break
        meta = dict(name=name, kind=kind, max_steps=kw["max_steps"], noise_sorting=kw["noise_sorting"], balesize=kw["balesize"],
                    ctor_seed=int(seed), policy=policy, masking=bool(masking), check_overflow=bool(check_overflow),
                    numpy=np.__version__, config_overrides={}, generator="oracle/gen_golden.py --trace from the imported reference")
        out = _pack(rows, meta)
        for k, v in ref_harness.ledgers(env).items():
            out["ledger_" + k] = v
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:48s} rows={len(out['op']):4d} press_log={len(out['ledger_press_log'])} "
              f"bales={sum(len(out['ledger_bales_' + m]) for m in 'ABCDE')}  {os.path.getsize(path)/1024:7.1f} KiB")


def gen_random_mode():
    """Env_3_Monolith.step(action=None, mode='random') (env_monolith.py:152-164, 245-253): actions drawn from the
    process-global np.random that reset(seed) seeds (env_super.py:177); without masking the press action is sanitised
    AFTER sort_material.  Same row format as the step traces; `arg` of a step row = info["action"]."""
    cls = ref_harness.load()["mono"]
    for name, masking, noise, seeds, seg in (("randmode_mono_n0_masked_s5_s31", True, 0.0, [5, 31], 205),
                                             ("randmode_mono_n5_unmasked_s6_s12", False, 0.05, [6, 12], 205)):
        env = cls(max_steps=200, seed=seeds[0], noise_sorting=noise, balesize=200)
        rows = {k: [] for k in ("op", "arg", "sort_mode", "flags", "obs", "reward", "terminated", "mask", "ints", "dbls", "rng")}

        def record(op, arg, obs, rew, term):
            I, Dd, R = ref_harness.snapshot(env)
            for k, v in (("op", op), ("arg", arg), ("sort_mode", -1), ("flags", 0), ("obs", np.asarray(obs, dtype=np.float32)),
                         ("reward", float(rew)), ("terminated", int(term)), ("mask", np.asarray(env.action_masks(), dtype=np.uint8)),
                         ("ints", I), ("dbls", Dd), ("rng", R)):
                rows[k].append(v)

        for seed in seeds:
            obs, _ = env.reset(seed=seed)
            record(1, seed, obs, 0.0, 0)
            for _ in range(seg):
                obs, rew, term, trunc, info = env.step(action=None, mode="random", use_action_masking=masking)
                record(0, int(info["action"]), obs, rew, term)
        meta = dict(name=name, kind="mono", max_steps=200, noise_sorting=noise, balesize=200, ctor_seed=int(seeds[0]),
                    policy="mode='random' (global np.random)", masking=bool(masking), check_overflow=False,
                    numpy=np.__version__, config_overrides={}, generator="oracle/gen_golden.py --random-mode from the imported reference")
        out = _pack(rows, meta)
        for k, v in ref_harness.ledgers(env).items():
            out["ledger_" + k] = v
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:48s} rows={len(out['op']):4d}  {os.path.getsize(path)/1024:7.1f} KiB")


if __name__ == "__main__" and "--random-mode" in sys.argv:
    gen_random_mode()
if __name__ == "__main__" and "--model" in sys.argv:
    gen_model_fallback()
if __name__ == "__main__" and "--trace" in sys.argv:
    gen_trace()


def gen_agents():
    """Env_3_Monolith.step(action=None, ...) with AGENTS assigned, scripted and recording (oracle/ref_harness.py):
      agents_mono_monoagent_*  a stored mono_agent (env_monolith.py:144-150): `agent_obs` f32[rows, 29] and `agent_mask`
                               u8[rows, 22] are what the reference handed predict() inside each step - get_obs() AFTER
                               the step's flow update - `arg` the action it answered (info["action"]);
      agents_mono_model_*      mode='model' (env_monolith.py:186-221) with a sort_agent and / or a press_agent, maskable
                               or not: `sort_agent_obs` f32[rows, 13], `press_agent_obs` f32[rows, 16], `press_agent_mask`
                               u8[rows, 11] with `press_agent_has_mask` u8[rows]; a part no agent decides is the env's own
                               rng_sorting / rng_pressing draw.
    Same row format as the step traces otherwise; the agents' answers are a function of a fixture PRNG and the mask."""
    cls = ref_harness.load()["mono"]
    cases = (
        # name, noise, seeds, segment, masking, mono | (sort agent?, press agent: None | "maskable" | "plain"), respect mask?
        ("agents_mono_monoagent_n0_s3_s21", 0.0, [3, 21], 210, True, "mono", True),
        ("agents_mono_monoagent_n5_ignoremask_s4_s9", 0.05, [4, 9], 210, True, "mono", False),
        ("agents_mono_model_both_maskable_n0_s6_s15", 0.0, [6, 15], 210, True, (True, "maskable"), True),
        ("agents_mono_model_both_plain_n5_s8_s2", 0.05, [8, 2], 210, True, (True, "plain"), False),
        ("agents_mono_model_sortonly_n0_s10_s33", 0.0, [10, 33], 210, True, (True, None), True),
        ("agents_mono_model_pressonly_n5_unmasked_s12_s5", 0.05, [12, 5], 210, False, (False, "maskable"), False),
    )
    for name, noise, seeds, seg, masking, who, respect in cases:
        env = cls(max_steps=200, seed=seeds[0], noise_sorting=noise, balesize=200)
        prng = np.random.default_rng(sum(name.encode()) * 15485863)
        n_act = 22 if who == "mono" else 11

        def decide(mask, prng=prng, n_act=n_act, respect=respect):
            if mask is not None and respect:
                return prng.choice(np.flatnonzero(mask))
            return prng.integers(0, n_act)       # an agent that ignores (or never sees) the mask

        sort_agent = press_agent = mono_agent = None
        if who == "mono":
            mono_agent = ref_harness.ScriptedMonoAgent(decide)
            env.set_agents(mono_agent=mono_agent)
        else:
            if who[0]:
                sort_agent = ref_harness.ScriptedSortAgent()
            if who[1] is not None:
                press_agent = (ref_harness.MaskableScriptedPressAgent if who[1] == "maskable" else ref_harness.ScriptedPressAgent)(decide)
            env.set_agents(sort_agent=sort_agent, press_agent=press_agent)
        rows = {k: [] for k in ("op", "arg", "sort_mode", "flags", "obs", "reward", "terminated", "mask", "ints", "dbls", "rng")}
        extra = {k: [] for k in ("agent_obs", "agent_mask", "sort_agent_obs", "press_agent_obs", "press_agent_mask", "press_agent_has_mask")}

        def record(op, arg, obs, rew, term):
            I, Dd, R = ref_harness.snapshot(env)
            for k, v in (("op", op), ("arg", arg), ("sort_mode", -1), ("flags", 0), ("obs", np.asarray(obs, dtype=np.float32)),
                         ("reward", float(rew)), ("terminated", int(term)), ("mask", np.asarray(env.action_masks(), dtype=np.uint8)),
                         ("ints", I), ("dbls", Dd), ("rng", R)):
                rows[k].append(v)

        def record_agents(step):
            z13, z16, z11 = np.zeros(13, np.float32), np.zeros(16, np.float32), np.zeros(11, np.uint8)
            extra["agent_obs"].append(mono_agent.last_obs if step and mono_agent else np.zeros(29, np.float32))
            extra["agent_mask"].append(mono_agent.last_mask if step and mono_agent else np.zeros(22, np.uint8))
            extra["sort_agent_obs"].append(sort_agent.last_obs if step and sort_agent else z13)
            extra["press_agent_obs"].append(press_agent.last_obs if step and press_agent else z16)
            has = bool(step and press_agent and press_agent.last_mask is not None)
            extra["press_agent_mask"].append(press_agent.last_mask if has else z11)
            extra["press_agent_has_mask"].append(int(has))

        for seed in seeds:
            obs, _ = env.reset(seed=seed)
            record(1, seed, obs, 0.0, 0)
            record_agents(False)
            for _ in range(seg):
                if sort_agent is not None:
                    sort_agent.next_mode = int(prng.integers(0, 2))
                if who == "mono":
                    obs, rew, term, trunc, info = env.step()                     # every argument at its default
                else:
                    obs, rew, term, trunc, info = env.step(action=None, mode="model", use_action_masking=masking)
                record(0, int(info["action"]), obs, rew, term)
                record_agents(True)
        meta = dict(name=name, kind="mono", max_steps=200, noise_sorting=noise, balesize=200, ctor_seed=int(seeds[0]),
                    policy="stored mono_agent" if who == "mono" else "mode='model' with agents", masking=bool(masking),
                    agents=dict(mono=who == "mono", sort=bool(who != "mono" and who[0]),
                                press=(None if who == "mono" else who[1])),
                    check_overflow=False, numpy=np.__version__, config_overrides={},
                    generator="oracle/gen_golden.py --agents from the imported reference")
        out = _pack(rows, meta)
        out["agent_obs"] = np.asarray(extra["agent_obs"], dtype=np.float32)
        out["agent_mask"] = np.asarray(extra["agent_mask"], dtype=np.uint8)
        out["sort_agent_obs"] = np.asarray(extra["sort_agent_obs"], dtype=np.float32)
        out["press_agent_obs"] = np.asarray(extra["press_agent_obs"], dtype=np.float32)
        out["press_agent_mask"] = np.asarray(extra["press_agent_mask"], dtype=np.uint8)
        out["press_agent_has_mask"] = np.asarray(extra["press_agent_has_mask"], dtype=np.uint8)
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:52s} rows={len(out['op']):4d}  {os.path.getsize(path)/1024:7.1f} KiB")


CONFIG_D = {"pressing_station": {"press_times": {1: 1, 2: 1}, "container_capacity": 260}}


def gen_overflow_info():
    """step(..., check_overflow=True) until the overflow termination, several episodes per fixture, with the `info`
    dict the reference returned (env_monolith.py:264-268, env_2_press.py:144-150, env_1_sort.py:131-139):
    `info_overflow` u8, `info_material` i8 (0..4 = A..E, -1 = key absent), `info_action` i16 (the "action" value;
    Env_1's overflow case reports the press job [press_id, material_id] it sampled itself, stored as its discrete
    code (press_id - 1) * 5 + material_id + 1, [0, None] as 0).  Under CONFIG_D (1-step presses, capacity 260) a policy
    that presses the fullest container but never material X makes X the one that overflows, E included."""
    classes = ref_harness.load()
    cases = (
        ("info_mono_n5_noop_s1_s2_s7", "mono", dict(max_steps=200, noise_sorting=0.05, balesize=200), [1, 2, 7], "noop", None, None),
        ("info_mono_cfgD_n0_avoidE_s3_s4", "mono", dict(max_steps=200, noise_sorting=0.0, balesize=20), [3, 4], "avoid", 4, CONFIG_D),
        ("info_mono_cfgD_n5_avoidC_s5_s6", "mono", dict(max_steps=200, noise_sorting=0.05, balesize=20), [5, 6], "avoid", 2, CONFIG_D),
        ("info_press_cfgD_n0_avoidE_s8_s9", "press", dict(max_steps=200, noise_sorting=0.0, balesize=20), [8, 9], "avoid", 4, CONFIG_D),
        ("info_press_cfgD_n5_avoidD_s10_s11", "press", dict(max_steps=200, noise_sorting=0.05, balesize=20), [10, 11], "avoid", 3, CONFIG_D),
        ("info_press_n0_maxsteps_s12", "press", dict(max_steps=5, noise_sorting=0.0, balesize=200), [12, 13], "noop", None, None),
        ("info_sort_n5_s14_s15_s16", "sort", dict(max_steps=200, noise_sorting=0.05, balesize=200), [14, 15, 16], "noop", None, None),
        ("info_sort_cfgD_n0_s17_s18", "sort", dict(max_steps=200, noise_sorting=0.0, balesize=250), [17, 18], "noop", None, CONFIG_D),
    )
    for name, kind, kw, seeds, policy, avoid, overrides in cases:
        env = _construct(classes[kind], seeds[0], kw, overrides)
        agent = None
        if kind == "press":
            agent = ref_harness.ScriptedSortAgent()
            env.set_agents(sort_agent=agent)
        prng = np.random.default_rng(sum(name.encode()) * 32452843)
        rows = {k: [] for k in ("op", "arg", "sort_mode", "flags", "obs", "reward", "terminated", "mask", "ints", "dbls", "rng")}
        info_rows = []

        def record(op, arg, sm, flags, obs, rew, term, info):
            I, Dd, R = ref_harness.snapshot(env)
            for k, v in (("op", op), ("arg", arg), ("sort_mode", sm), ("flags", flags), ("obs", np.asarray(obs, dtype=np.float32)),
                         ("reward", float(rew)), ("terminated", int(term)), ("mask", np.asarray(env.action_masks(), dtype=np.uint8)),
                         ("ints", I), ("dbls", Dd), ("rng", R)):
                rows[k].append(v)
            a = info.get("action", 0)
            if isinstance(a, (list, tuple)):                       # Env_1's overflow case: the sampled press job
                a = 0 if not a[0] else (int(a[0]) - 1) * 5 + int(a[1]) + 1
            assert set(info) <= {"overflow", "overflow_material", "action"}
            info_rows.append((int(bool(info.get("overflow", False))),
                              ref_harness.MATERIALS.index(info["overflow_material"]) if "overflow_material" in info else -1, int(a)))

        for seed in seeds:
            obs, _ = env.reset(seed=seed)
            record(1, seed, -1, 0, obs, 0.0, 0, {})
            for t in range(kw["max_steps"]):
                sm = int(prng.integers(0, 2))
                if policy == "noop":
                    a = sm if kind == "sort" else (11 * sm if kind == "mono" else 0)
                else:  # press the fullest container except `avoid`, if the mask allows it
                    mask = np.asarray(env.action_masks(), dtype=bool)[:11]
                    cm = env.container_materials
                    lvl = [cm[m] + cm.get(m + "_False", 0) for m in "ABCDE"]
                    order = [m for m in np.argsort(lvl)[::-1] if m != avoid]
                    p = 0
                    for m in order:
                        cand = [c for c in (1 + m, 6 + m) if mask[c]]
                        if cand:
                            p = cand[0]
                            break
                    a = p + 11 * sm if kind == "mono" else p
                if kind == "press":
                    agent.next_mode = sm
                obs, rew, term, trunc, info = env.step(a, use_action_masking=True, check_overflow=True)
                record(0, a, sm if kind == "press" else -1, STEP_CHECK_OVERFLOW, obs, rew, term, info)
                if term:
                    break
        meta = dict(name=name, kind=kind, max_steps=kw["max_steps"], noise_sorting=kw["noise_sorting"], balesize=kw["balesize"],
                    ctor_seed=int(seeds[0]), policy=policy if avoid is None else f"fullest container except {'ABCDE'[avoid]}",
                    masking=True, check_overflow=True, numpy=np.__version__, config_overrides=overrides or {},
                    generator="oracle/gen_golden.py --overflow-info from the imported reference")
        out = _pack(rows, meta)
        inf = np.asarray(info_rows, dtype=np.int64)
        out["info_overflow"], out["info_material"], out["info_action"] = inf[:, 0].astype(np.uint8), inf[:, 1].astype(np.int8), inf[:, 2].astype(np.int16)
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        ends = [(int(m), int(t)) for m, t, o in zip(out["info_material"], out["terminated"], out["op"]) if t]
        print(f"{name:44s} rows={len(out['op']):4d} episode ends (material, terminated)={ends}  {os.path.getsize(path)/1024:6.1f} KiB")


if __name__ == "__main__" and "--agents" in sys.argv:
    gen_agents()
if __name__ == "__main__" and "--overflow-info" in sys.argv:
    gen_overflow_info()
