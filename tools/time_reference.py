"""Time the REFERENCE's own NumPy env on this container's host cores (SURVEY 8d CPU baseline (1), BASELINE.json
configs[0]: Env_1_Sorting, 1 env, random actions, 10 000 steps) and write profiles/r03/reference_cpu.json.

    python tools/time_reference.py          # build container only: /root/reference does not exist on the GPU box

The reference is imported through oracle/ref_harness.py (test infrastructure; three arithmetic-irrelevant imports
replaced by inert placeholders) and only ever runs here.  bench.py quotes the resulting file, with its provenance,
under cpu_baseline.reference_numpy; it never runs the reference itself.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_env(kind: str, steps: int, seed: int = 42, repeats: int = 3):
    """`steps` steps of one env under a masked-uniform random policy with reset on termination (max_steps 200, noise 0,
    balesize 200: SURVEY 8d "All"); best of `repeats` (the shortest run is the least disturbed one)."""
    cls = ref_harness.load()[kind]
    best = None
    for r in range(repeats):
        env = cls(max_steps=200, seed=seed, noise_sorting=0.0, balesize=200)
        prng = np.random.default_rng(2024 + r)
        with contextlib.redirect_stdout(io.StringIO()):
            env.reset(seed=seed)
            t0 = time.perf_counter()
            for _ in range(steps):
                if kind == "sort":
                    a = int(prng.integers(0, 2))
                else:
                    a = int(prng.choice(np.flatnonzero(env.action_masks())))
                _, _, term, _, _ = env.step(a)
                if term:
                    env.reset()
            dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"steps": steps, "seconds": best, "env_steps_per_s": steps / best}


def main():
    if not ref_harness.available():
        raise SystemExit("reference checkout not available: this script only runs in the build container")
    out = {
        "what": "the reference's own Python/NumPy envs, one process, one core, masked-uniform random actions, reset on "
                "termination, max_steps 200, noise 0, balesize 200",
        "where": "build container (never the GPU box: the reference cannot travel)",
        "cpu_model": cpu_model(), "cpu_count": os.cpu_count(), "cores_used": 1,
        "python": platform.python_version(), "numpy": np.__version__,
        "configs": {
            "configs[0] Env_1_Sorting, 1 env, 10k steps": time_env("sort", 10_000),
            "Env_2_Pressing, 1 env, 10k steps": time_env("press", 10_000),
            "Env_3_Monolith, 1 env, 10k steps": time_env("mono", 10_000),
        },
        "generated_by": "tools/time_reference.py",
    }
    path = os.path.join(ROOT, "profiles", "r03", "reference_cpu.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
