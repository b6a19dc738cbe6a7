"""Per-role section timeline of k_rollout_ring from the diagnostic build (-DMSE_TIMELINE).

    hipcc ... -DMSE_TIMELINE marl-sortingenv_amd/csrc/mse_lib.hip -o build/abl/libmse_timeline.so
    python tools/timeline.py [lib] [kind] [envs]

Prints, per role (dynamics / observer / RNG waves), the mean s_memtime ticks per step spent in each
section, and the share of the step."""
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
noise = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
launches = 13
env = M.BatchedSortingEnv(kind=kind, num_envs=n, device=0, base_seed=0, max_steps=200, noise_sorting=noise,
                          balesize=200, library=lib)
buf = env.alloc_rollout(K)
L = env.L
L.mse_debug_timeline.argtypes = [C.POINTER(C.c_uint64 * 32)]
out = (C.c_uint64 * 32)()
for _ in range(2):
    env.rollout(K, buffers=buf)
L.mse_debug_timeline(C.byref(out))  # clear the warm-up
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches):
    env.rollout(K, buffers=buf)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / launches
assert L.mse_debug_timeline(C.byref(out)) == 0
waves = 4  # workgroup 0 only
names = {
    0: ["policy", "flow+accuracy", "sort_material w/o loops", "press", "snapshot/levels", "mask+snap write", "barrier wait", "draw loops"],
    1: ["barrier wait", "snap read+observe", "stage+store", "", "", "", "", ""],
    2: ["produce", "barrier wait", "", "trip count M (sum, not ticks)", "outputs produced by the wave's 64 lanes (sum, not ticks)", "", "", ""],
}
print(f"{kind} {n} envs noise {noise}: {ms * 1e3:.1f} us per launch of {K} steps = {ms * 1e3 / K:.2f} us/step (instrumented build)")
ev = [out[24 + k] / (4 * launches) for k in range(6)]
print("  launch edges (dynamics wave, ticks per launch): " + ", ".join(
    f"{nm} {v:.0f}" for nm, v in zip(["tables->LDS", "state load", "ring priming wait", "step loop", "jump-ahead", "state store"], ev)))
for role, label in ((0, "dynamics"), (1, "observer"), (2, "rng")):
    vals = [out[role * 8 + k] / (waves * launches * K) for k in range(8)]
    tot = sum(vals)
    print(f"  {label:9s} ticks/step total {tot:8.1f}: " +
          ", ".join(f"{nm} {v:.1f} ({100 * v / tot:.0f}%)" for nm, v in zip(names[role], vals) if nm))
