"""Per-role section timeline of k_rollout_policy_roles from the diagnostic build (-DMSE_TIMELINE): mean shader cycles per
step and wave in each section of the actor, critic and RNG waves (workgroup 0 only).

    hipcc ... -DMSE_TIMELINE marl-sortingenv_amd/csrc/mse_lib.hip marl-sortingenv_amd/csrc/mse_policy.hip -o build/abl/libmse_timeline.so
    python tools/timeline_roles.py [lib] [kind] [envs] [K] [rollout_pipeline]"""
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
pipeline = int(sys.argv[5]) if len(sys.argv) > 5 else 0
launches = 9
env = M.BatchedSortingEnv(kind=kind, num_envs=n, device=0, base_seed=0, max_steps=200, noise_sorting=0.0,
                          balesize=200, library=lib, rollout_pipeline=pipeline)
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
names = {
    0: ["actor + sampling (tile 0)", "flow+accuracy", "sort_material w/o loops", "press", "snapshot/levels",
        "observe, mask, stores, row post", "barrier + tile-1 wait", "draw loops"],
    1: ["barrier wait", "tile -> registers", "row stores", "value network", "actor + sampling (tile 1)", "", "", ""],
    2: ["produce", "barrier wait", "", "trip count M (sum, not cycles)", "outputs produced (sum, not cycles)", "", "", ""],
}
print(f"{kind} {n} envs, pipeline {pipeline}: {ms * 1e3:.1f} us per launch of {K} steps = {ms * 1e3 / K:.2f} us/step "
      f"(instrumented build), {n * K / ms / 1e6:.2f} G env-steps/s")
names[3] = ["operand split", "layer 1 (LDS + MFMA)", "tanh + split", "layer 2", "tanh + split", "head", "masked softmax + sample", ""]
for role, label in ((0, "actor"), (1, "critic"), (2, "rng"), (3, "actor's tile 0, by stage (-DMSE_TIMELINE=3)")):
    vals = [out[role * 8 + k] / (4 * launches * K) for k in range(8)]
    tot = sum(v for v, nm in zip(vals, names[role]) if nm and "sum" not in nm)
    if tot == 0:
        continue
    print(f"  {label:7s} cycles/step {tot:8.1f}: " +
          ", ".join(f"{nm} {v:.1f}" + (f" ({100 * v / tot:.0f}%)" if "sum" not in nm else "") for nm, v in zip(names[role], vals) if nm))
