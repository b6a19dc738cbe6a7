"""In-kernel clock and cycles per step of k_rollout_ring under SUSTAINED load, from a -DMSE_CLOCK_PROBE build
(MI355X_MICROARCH.md "DVFS give-back" (6): clock = d s_memtime / d s_memrealtime x 100 MHz).

    hipcc ... -DMSE_CLOCK_PROBE [-DMSE_ABL_...] ... -o build/abl/libmse_clk.so
    python tools/clock_probe.py LIB [K] [noise] [seconds]

One lane of workgroup 0 stamps the start and the end of every launch: a handful of scalar instructions per launch,
nothing in the step loop."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import marl_sortingenv_amd as M  # noqa: E402

lib = os.path.abspath(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
env = M.BatchedSortingEnv(kind="mono", num_envs=65536, device=0, base_seed=0, max_steps=200, noise_sorting=noise,
                          balesize=200, library=lib)
buf = env.alloc_rollout(K)
L = env.L
L.mse_debug_clock.argtypes = [C.POINTER(C.c_uint64 * 3)]
out = (C.c_uint64 * 3)()
t0 = time.perf_counter()
while time.perf_counter() - t0 < seconds:            # bring the chip to its sustained state
    for _ in range(200):
        env.rollout(K, buffers=buf)
    torch.cuda.synchronize()
L.mse_debug_clock(C.byref(out))
n = 2000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    env.rollout(K, buffers=buf)
e1.record()
torch.cuda.synchronize()
assert L.mse_debug_clock(C.byref(out)) == 0 and out[2] == n
cyc, ticks = out[0] / n, out[1] / n
us = e0.elapsed_time(e1) * 1e3 / n
print(f"{os.path.basename(lib):28s} K={K} noise={noise}: {us:7.2f} us per launch (events), in-kernel {ticks / 100:7.2f} us, "
      f"{cyc:9.0f} cycles = {cyc / K:7.1f} per step, clock {cyc / ticks * 100:6.0f} MHz")

# per-workgroup start / end of the LAST launch (100 MHz ticks): how the kernel's duration relates to a workgroup's own
import numpy as np  # noqa: E402

L.mse_debug_wg_probe.argtypes = [C.POINTER(C.c_uint64 * 3072)]
wg = (C.c_uint64 * 3072)()
assert L.mse_debug_wg_probe(C.byref(wg)) == 0
a = np.array(wg[:3 * 256], dtype=np.int64).reshape(256, 3)
st, en, xcc = (a[:, 0] - a[:, 0].min()) / 100.0, (a[:, 1] - a[:, 0].min()) / 100.0, a[:, 2]
life = en - st
q = lambda v: "min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(v, [0, 10, 50, 90, 100]))
print("  workgroup starts after the first (us): " + q(st))
print("  workgroup ends   after the first start: " + q(en))
print("  workgroup lifetimes (us):               " + q(life))
print("  first start -> last end %.2f us; events say %.2f us per launch" % (en.max(), us))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"    XCC {x}: {int(m.sum())} workgroups, start median {np.median(st[m]):.2f}, end median {np.median(en[m]):.2f}, lifetime median {np.median(life[m]):.2f}")
