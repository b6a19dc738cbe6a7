"""Phase timeline of k_rollout_policy from the diagnostic build (-DMSE_TIMELINE=2): mean s_memtime ticks per step
and wave spent storing the row, in the policy forward, in the env transition and in the per-env stores
(workgroup 0 only).

    python tools/timeline_policy.py [lib] [kind] [envs] [K]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import marl_sortingenv_amd as M  # noqa: E402

lib = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build/abl/libmse_timeline.so"))
kind = sys.argv[2] if len(sys.argv) > 2 else "mono"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
K = int(sys.argv[4]) if len(sys.argv) > 4 else 16
launches = 9
env = M.BatchedSortingEnv(kind=kind, num_envs=n, device=0, base_seed=0, max_steps=200, noise_sorting=0.0,
                          balesize=200, library=lib)
pol = M.MlpPolicy.random_init(env.obs_dim, env.num_actions, seed=1, device=0, library=lib)
fused = M.FusedPolicyRollout(env, pol, K, seed=3)
L = env.L
L.mse_debug_timeline.argtypes = [C.POINTER(C.c_uint64 * 32)]
out = (C.c_uint64 * 32)()
for _ in range(2):
    fused.collect()
L.mse_debug_timeline(C.byref(out))  # clear the warm-up
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches):
    fused.collect()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / launches
assert L.mse_debug_timeline(C.byref(out)) == 0
waves = 8 if n > 32 * 8 * 256 or pol.precision == "f16x3" else 4
names = ["row stores", "policy forward", "sorting agent", "env transition", "mask + per-env stores"]
vals = [out[k] / (waves * launches * K) for k in range(5)]
tot = sum(vals)
print(f"{kind} {n} envs, {pol.precision}: {ms * 1e3:.1f} us per launch of {K} steps = {ms * 1e3 / K:.2f} us/step "
      f"(instrumented build), {n * K / ms / 1e6:.2f} G env-steps/s")
print(f"  per wave and step {tot:.0f} ticks: " + ", ".join(f"{nm} {v:.0f} ({100 * v / tot:.0f}%)" for nm, v in zip(names, vals)))
