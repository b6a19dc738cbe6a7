"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/mse.h
declares, refuses to run without a device (no CPU fallback), and the host config logic."""
import ctypes as C
import os
import re

import pytest

import marl_sortingenv_amd as M
from marl_sortingenv_amd._lib import EXPORTS, MseConfigStruct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mse.h")).read()
    return sorted(set(re.findall(r"\b(mse_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(M.library_path()):
        M.build_library()
    L = M.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"libmse_hip.so does not export {name}"
    assert sorted(EXPORTS) == declared
    assert L.mse_version() == 200


def test_config_struct_matches_header_and_defaults():
    L = M.load_library()
    c = MseConfigStruct()
    assert L.mse_config_default(C.byref(c)) == 0
    assert c.struct_size == C.sizeof(MseConfigStruct)
    assert (c.press_time[0], c.press_time[1], c.container_capacity, c.bale_standard_size) == (12, 15, 700, 200)
    assert [c.pattern_ratio[0][m] for m in range(4)] == [0.40, 0.15, 0.35, 0.10]
    s = M.SortingEnvConfig().to_struct("press", max_steps=200, noise_sorting=0.0, balesize=150)
    assert (s.env_kind, s.max_steps, s.noise, s.bale_standard_size) == (2, 200, 0.0, 150)
    assert s.quality_threshold_r2[0] == 0.9


def test_no_device_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a device is present")
    L = M.load_library()
    c = M.SortingEnvConfig().to_struct("mono")
    h = C.c_void_p()
    rc = L.mse_create(C.byref(h), C.byref(c), 16, 0)
    assert rc == -3 and b"no CPU path" in L.mse_last_error()
    with pytest.raises(RuntimeError):
        M.BatchedSortingEnv(kind="mono", num_envs=4)


def test_create_rejects_bad_arguments():
    L = M.load_library()
    h = C.c_void_p()
    c = M.SortingEnvConfig().to_struct("mono")
    c.struct_size = 7
    assert L.mse_create(C.byref(h), C.byref(c), 16, 0) == -1
    c = M.SortingEnvConfig().to_struct("mono")
    assert L.mse_create(C.byref(h), C.byref(c), 0, 0) == -1
    c.max_steps = 70000
    assert L.mse_create(C.byref(h), C.byref(c), 4, 0) == -2


def test_config_from_yaml_roundtrip(tmp_path):
    y = tmp_path / "config.yml"
    y.write_text(
        "simulation: {input_occupancy_min: 60, input_occupancy_max: 80, input_batch_size: 100, steps_per_pattern: 20, input_history_length: 10}\n"
        "sorting_station: {baseline_accuracy: [0.7, 0.7, 0.8, 0.8], boost: 0.25, occupancy_reduction_factor: 0.2, noise: 0.01, stage_capacity: 100}\n"
        "pressing_station:\n  press_times: {1: 10, 2: 20}\n  container_capacity: 650\n  bale_standard_size: 180\n"
        "  bale_remainder_threshold: 0.5\n  bale_quality_thresholds: {A: 0.9, B: 0.85, C: 0.9, D: 0.9}\n"
        "rewards:\n  sorting: {purity_threshold_theta: 0.8, tanh_temperature: 0.5}\n"
        "  pressing: {overflow_penalty_catastrophic: -1.0, overflow_penalty_severe: -0.5, overflow_penalty_mild: -0.2, bale_efficiency_factor: 1, max_state_reward: 0.5}\n"
        "  overflow_termination_penalty: -10.0\n")
    cfg = M.SortingEnvConfig.from_yaml(str(y))
    s = cfg.to_struct("mono")
    assert (s.press_time[0], s.press_time[1], s.container_capacity, s.bale_standard_size) == (10, 20, 650, 180)
    assert s.baseline_accuracy[2] == 0.8 and s.boost == 0.25 and s.quality_threshold[1] == 0.85
    assert s.noise == 0.01  # noise_sorting=None falls back to the file (env_super.py:71)
    assert cfg.to_struct("mono", noise_sorting=0.05).noise == 0.05
