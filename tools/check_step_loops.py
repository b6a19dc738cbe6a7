"""Lists what the rollout kernels' step loops do to memory, from a `hipcc -S` listing of mse_lib.hip:

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-fast-math -I include -S --offload-device-only \
          marl-sortingenv_amd/csrc/mse_lib.hip -o /tmp/mse_lib.s
    python tools/check_step_loops.py /tmp/mse_lib.s

For every depth-1 loop of k_rollout / k_rollout_ring / k_rollout_policy it counts global / scratch loads, vmcnt waits and
scalar loads between the loop header and the function's epilogue.  The step loops must hold no global or scratch load
and no vmcnt wait: such a wait also covers the wave's own output stores (the one global load round 2
found there - a per-lane selection among kernel arguments - cost 3 %).  Exit status 1 if one is found."""
import re
import sys

path = sys.argv[1]
lines = open(path).read().split("\n")
KERNELS = ["_Z9k_rolloutILi", "_Z14k_rollout_ringILi", "_Z16k_rollout_policyILi"]
bad = 0
i = 0
while i < len(lines):
    l = lines[i]
    if l.startswith("_Z") and any(l.startswith(k) for k in KERNELS) and l.split(";")[0].strip().endswith(":"):
        name = l.split(":")[0]
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[i:end]
        # the step loop(s): depth-1 loop headers that have child loops (the draw / production loops) or are long
        heads = [j for j, b in enumerate(body) if "Loop Header: Depth=1" in b and "Inner Loop" not in b]
        # the loop's extent: up to the last line that says it belongs to that header
        for h in heads:
            label = body[h].split(":")[0].lstrip(".")
            pat = re.compile(r"(Header=|Loop )" + re.escape(label[1:]) + r"\b")
            members = [j for j, b in enumerate(body) if pat.search(b)]
            last = max(members) if members else h
            # blocks are contiguous in practice; scan header .. last member block end
            stop = next((j for j in range(last + 1, len(body)) if body[j].startswith(".LBB") or body[j].startswith("; %bb.")), len(body))
            seg = body[h:stop]
            ops = [s.split(";")[0].strip().split()[0] for s in seg if s.startswith("\t") and s.split(";")[0].strip() and not s.strip().startswith(".")]
            n_gl = sum(1 for o in ops if o.startswith(("global_load", "flat_load", "buffer_load")))
            n_sc = sum(1 for o in ops if o.startswith("scratch_"))
            n_vm = sum(1 for s in seg if "s_waitcnt" in s and "vmcnt" in s)
            n_sl = sum(1 for o in ops if o.startswith("s_load"))
            n_st = sum(1 for o in ops if o.startswith("global_store"))
            flag = "  <-- memory wait inside the step loop" if (n_gl or n_sc or n_vm) else ""
            print(f"{name[:58]:58s} loop {label:10s} {len(ops):5d} instr: global loads {n_gl}, scratch {n_sc}, vmcnt waits {n_vm}, "
                  f"s_load {n_sl}, global stores {n_st}{flag}")
            bad += 1 if flag else 0
        i = end
    i += 1
sys.exit(1 if bad else 0)
