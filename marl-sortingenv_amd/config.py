"""Host-side configuration: the reference's config.yml schema -> the POD struct of kernel constants.

Mirrors what Env_Super.__init__ reads (reference src/envs_train/env_super.py:25-137) without
shipping the reference's file: defaults below are config.yml's values; `from_yaml` accepts a
user's config.yml in the same schema.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field

from ._lib import MSE_ENV_MONO, MSE_ENV_PRESS, MSE_ENV_SORT, MseConfigStruct, load_library

KIND_BY_NAME = {"sort": MSE_ENV_SORT, "press": MSE_ENV_PRESS, "mono": MSE_ENV_MONO}
NAME_BY_KIND = {v: k for k, v in KIND_BY_NAME.items()}
OBS_DIM = {"sort": 13, "press": 16, "mono": 29}
NUM_ACTIONS = {"sort": 2, "press": 11, "mono": 22}


@dataclass
class SortingEnvConfig:
    # simulation
    input_occupancy_min: int = 60
    input_occupancy_max: int = 80
    input_batch_size: int = 100
    steps_per_pattern: int = 20
    # sorting_station
    baseline_accuracy: tuple = (0.75, 0.75, 0.75, 0.75)
    boost: float = 0.5
    noise: float = 0.05
    stage_capacity: int = 100
    # pressing_station
    press_times: tuple = (12, 15)
    container_capacity: int = 700
    bale_standard_size: int = 200
    bale_remainder_threshold: float = 0.5
    bale_quality_thresholds: tuple = (0.9, 0.9, 0.9, 0.9)
    # rewards
    purity_threshold_theta: float = 0.80
    tanh_temperature: float = 0.5
    overflow_penalty_catastrophic: float = -1.0
    overflow_penalty_severe: float = -0.5
    overflow_penalty_mild: float = -0.2
    bale_efficiency_factor: float = 1.0
    max_state_reward: float = 0.5
    overflow_termination_penalty: float = -10.0
    # seasonal patterns 1 and 2 in material order A, B, C, D
    pattern_ratios: tuple = ((0.40, 0.15, 0.35, 0.10), (0.15, 0.40, 0.10, 0.35))
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_yaml(cls, path: str) -> "SortingEnvConfig":
        import yaml

        with open(path, "r") as f:
            return cls.from_dict(yaml.safe_load(f))

    @classmethod
    def from_dict(cls, y: dict) -> "SortingEnvConfig":
        """From a mapping with config.yml's schema (simulation / sorting_station / pressing_station / rewards)."""
        sim, srt, prs, rew = y["simulation"], y["sorting_station"], y["pressing_station"], y["rewards"]
        q = prs["bale_quality_thresholds"]
        return cls(
            input_occupancy_min=sim["input_occupancy_min"], input_occupancy_max=sim["input_occupancy_max"],
            input_batch_size=sim["input_batch_size"], steps_per_pattern=sim["steps_per_pattern"],
            baseline_accuracy=tuple(srt["baseline_accuracy"]), boost=srt["boost"], noise=srt["noise"],
            stage_capacity=srt["stage_capacity"],
            press_times=(prs["press_times"][1], prs["press_times"][2]),
            container_capacity=prs["container_capacity"], bale_standard_size=prs["bale_standard_size"],
            bale_remainder_threshold=prs["bale_remainder_threshold"],
            bale_quality_thresholds=(q["A"], q["B"], q["C"], q["D"]),
            purity_threshold_theta=rew["sorting"]["purity_threshold_theta"],
            tanh_temperature=rew["sorting"]["tanh_temperature"],
            overflow_penalty_catastrophic=rew["pressing"]["overflow_penalty_catastrophic"],
            overflow_penalty_severe=rew["pressing"]["overflow_penalty_severe"],
            overflow_penalty_mild=rew["pressing"]["overflow_penalty_mild"],
            bale_efficiency_factor=rew["pressing"].get("bale_efficiency_factor", 0.5),
            max_state_reward=rew["pressing"]["max_state_reward"],
            overflow_termination_penalty=rew["overflow_termination_penalty"],
        )

    def to_dict(self) -> dict:
        """The same values in config.yml's schema (only the keys the step path reads)."""
        q = self.bale_quality_thresholds
        return {
            "simulation": {"input_occupancy_min": self.input_occupancy_min, "input_occupancy_max": self.input_occupancy_max,
                           "input_batch_size": self.input_batch_size, "steps_per_pattern": self.steps_per_pattern},
            "sorting_station": {"baseline_accuracy": list(self.baseline_accuracy), "boost": self.boost, "noise": self.noise,
                                "stage_capacity": self.stage_capacity},
            "pressing_station": {"press_times": {1: self.press_times[0], 2: self.press_times[1]},
                                 "container_capacity": self.container_capacity,
                                 "bale_standard_size": self.bale_standard_size,
                                 "bale_remainder_threshold": self.bale_remainder_threshold,
                                 "bale_quality_thresholds": {"A": q[0], "B": q[1], "C": q[2], "D": q[3]}},
            "rewards": {"sorting": {"purity_threshold_theta": self.purity_threshold_theta,
                                    "tanh_temperature": self.tanh_temperature},
                        "pressing": {"overflow_penalty_catastrophic": self.overflow_penalty_catastrophic,
                                     "overflow_penalty_severe": self.overflow_penalty_severe,
                                     "overflow_penalty_mild": self.overflow_penalty_mild,
                                     "bale_efficiency_factor": self.bale_efficiency_factor,
                                     "max_state_reward": self.max_state_reward},
                        "overflow_termination_penalty": self.overflow_termination_penalty},
        }

    def with_overrides(self, overrides: dict) -> "SortingEnvConfig":
        """A copy with a nested dict in config.yml's schema merged over this config."""
        def merge(a, b):
            for k, v in b.items():
                if isinstance(v, dict) and isinstance(a.get(k), dict):
                    merge(a[k], v)
                else:
                    a[k] = v
            return a

        def intkeys(d):  # JSON turns press_times' integer keys into strings
            return {(int(k) if isinstance(k, str) and k.isdigit() else k): (intkeys(v) if isinstance(v, dict) else v)
                    for k, v in d.items()}

        return dataclasses.replace(type(self).from_dict(merge(self.to_dict(), intkeys(overrides))),
                                   pattern_ratios=self.pattern_ratios, extra=dict(self.extra))

    def to_struct(self, kind: str, max_steps: int = 50, noise_sorting=None, balesize=None,
                  auto_reset: bool = True, track_bales: bool = True, literal_choice: bool = False,
                  rollout_pipeline: int = 0) -> MseConfigStruct:
        import ctypes as C

        s = MseConfigStruct()
        load_library().mse_config_default(C.byref(s))
        s.env_kind = KIND_BY_NAME[kind]
        s.max_steps = int(max_steps)
        s.auto_reset = int(bool(auto_reset))
        s.track_bales = int(bool(track_bales))
        s.literal_choice = int(bool(literal_choice))
        s.rollout_pipeline = int(rollout_pipeline)
        s.input_batch_size = int(self.input_batch_size)
        s.steps_per_pattern = int(self.steps_per_pattern)
        for m in range(4):
            s.baseline_accuracy[m] = float(self.baseline_accuracy[m])
            s.quality_threshold[m] = float(self.bale_quality_thresholds[m])
            # purity of an empty container is python round(threshold, 2) (env_super.py:786-789)
            s.quality_threshold_r2[m] = float(round(float(self.bale_quality_thresholds[m]), 2))
        s.boost = float(self.boost)
        # ctor noise_sorting / balesize override the config (env_super.py:71, :87)
        s.noise = float(self.noise if noise_sorting is None else noise_sorting)
        s.stage_capacity = int(self.stage_capacity)
        s.press_time[0], s.press_time[1] = int(self.press_times[0]), int(self.press_times[1])
        s.container_capacity = int(self.container_capacity)
        s.bale_standard_size = int(self.bale_standard_size if balesize is None else balesize)
        s.bale_remainder_threshold = float(self.bale_remainder_threshold)
        s.purity_threshold_theta = float(self.purity_threshold_theta)
        s.tanh_temperature = float(self.tanh_temperature)
        s.overflow_penalty_catastrophic = float(self.overflow_penalty_catastrophic)
        s.overflow_penalty_severe = float(self.overflow_penalty_severe)
        s.overflow_penalty_mild = float(self.overflow_penalty_mild)
        s.bale_efficiency_factor = float(self.bale_efficiency_factor)
        s.max_state_reward = float(self.max_state_reward)
        s.overflow_termination_penalty = float(self.overflow_termination_penalty)
        for k in range(2):
            for m in range(4):
                s.pattern_ratio[k][m] = float(self.pattern_ratios[k][m])
        return s

    def replace(self, **kw) -> "SortingEnvConfig":
        return dataclasses.replace(self, **kw)
