"""Random config.yml's for the config-fuzz parity tests: every value the step path reads, inside the
ranges the engine accepts (include/mse.h mse_config).  Deterministic in the seed."""
import numpy as np


def fuzz_overrides(seed):
    """Nested dict in config.yml's schema + the ctor arguments (max_steps, noise_sorting, balesize)."""
    r = np.random.default_rng(1000 + seed)
    batch = int(r.choice([100, 100, 90, 77, 60, 120, 127, 40]))          # remainder-free and not; <= 127 (integer draw)
    cap = int(r.integers(300, 1200))
    bale = int(r.integers(60, min(cap, 400)))
    ov = {
        "simulation": {"input_batch_size": batch, "steps_per_pattern": int(r.integers(3, 40))},
        "sorting_station": {"baseline_accuracy": [round(float(x), 3) for x in r.uniform(0.35, 0.95, 4)],
                            "boost": round(float(r.uniform(0.0, 0.6)), 3),
                            "stage_capacity": int(r.choice([100, 128, 150, batch]))},
        "pressing_station": {"press_times": {1: int(r.integers(1, 30)), 2: int(r.integers(1, 30))},
                             "container_capacity": cap,
                             "bale_remainder_threshold": round(float(r.uniform(0.1, 0.9)), 2),
                             "bale_quality_thresholds": {m: round(float(r.uniform(0.5, 0.99)), int(r.choice([2, 3])))
                                                         for m in "ABCD"}},
        "rewards": {"sorting": {"purity_threshold_theta": round(float(r.uniform(0.5, 0.95)), 3),
                                "tanh_temperature": round(float(r.uniform(0.2, 1.5)), 3)},
                    "pressing": {"overflow_penalty_catastrophic": -round(float(r.uniform(0.5, 1.5)), 2),
                                 "overflow_penalty_severe": float(r.choice([0.0, -0.25, -0.5])),
                                 "overflow_penalty_mild": float(r.choice([0.0, -0.1, -0.2])),
                                 "bale_efficiency_factor": round(float(r.uniform(0.3, 1.2)), 2),
                                 "max_state_reward": round(float(r.uniform(0.1, 0.9)), 2)},
                    "overflow_termination_penalty": -round(float(r.uniform(2, 12)), 1)},
    }
    ctor = dict(max_steps=int(r.integers(20, 70)), noise_sorting=float(r.choice([0.0, 0.02, 0.1])), balesize=bale)
    return ov, ctor


def meta_for(kind, seed):
    ov, ctor = fuzz_overrides(seed)
    return dict(name=f"fuzz{seed}", kind=kind, ctor_seed=seed, config_overrides=ov, **ctor)
