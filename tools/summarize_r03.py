"""Summarise tools/prof_r03.sh's output directory: per-tag kernel stats (rocprofv3 --stats) and mean counter values per
dispatch of the dominant kernels, plus the HBM traffic entries bench.py's roofline.traffic reads (profiles/r03/traffic.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ("k_rollout_ring", "k_rollout_policy", "k_rollout<", "k_rollout_po")


def rows(pattern):
    out = []
    for p in glob.glob(pattern, recursive=True):
        with open(p) as f:
            out += list(csv.DictReader(f))
    return out


def main(d):
    out = {"dir": d, "tags": {}}
    traffic = {}
    for tag_dir in sorted(p for p in glob.glob(os.path.join(d, "*")) if os.path.isdir(p)):
        tag = os.path.basename(tag_dir)
        entry = {}
        stats = rows(os.path.join(tag_dir, "**", "*kernel_stats.csv"))
        if stats:
            entry["kernel_stats"] = [{k: r.get(k) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
                                     for r in stats if any(k in r.get("Name", "") for k in KERNELS)]
        ctr = rows(os.path.join(tag_dir, "**", "*counter_collection.csv"))
        if ctr:
            agg = defaultdict(lambda: defaultdict(list))
            for r in ctr:
                name = r["Kernel_Name"]
                if any(k in name for k in KERNELS):
                    agg[name.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            entry["counters_mean_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
            entry["dispatches"] = {k: max(len(v) for v in cs.values()) for k, cs in agg.items()}
        bench = os.path.join(d, tag + ".json")
        if os.path.exists(bench):
            try:
                line = [l for l in open(bench).read().splitlines() if l.startswith("{")][-1]
                b = json.loads(line)
                entry["bench"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "launch_ms": b["roofline"]["launch_ms"],
                                  "steps_per_launch": b["config"]["steps_per_launch"], "frac": b["roofline"]["frac"]}
            except Exception as e:  # noqa: BLE001
                entry["bench_error"] = str(e)
        out["tags"][tag] = entry
        for kind in ("fetch", "write"):
            if tag.startswith(kind + "_") and "counters_mean_per_dispatch" in entry:
                key = tag[len(kind) + 1:]
                for kname, cs in entry["counters_mean_per_dispatch"].items():
                    v = cs.get("FETCH_SIZE" if kind == "fetch" else "WRITE_SIZE")
                    if v is not None:
                        traffic.setdefault(key, {"kernel": kname})[kind + "_kb"] = v
    # valu_issue_ratio = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / waves per SIMD): the share of cycles in which a SIMD's vector
    # unit is issuing (both counters in quad-cycles; the multi-role kernel keeps 3 waves per SIMD, the learned-policy one 2)
    valu = {}
    for tag, entry in out["tags"].items():
        if tag.startswith("sq_"):
            for kname, cs in entry.get("counters_mean_per_dispatch", {}).items():
                if cs.get("SQ_WAVE_CYCLES"):
                    waves = 2.0 if "policy" in kname else 3.0
                    valu[tag[3:]] = cs["SQ_ACTIVE_INST_VALU"] / (cs["SQ_WAVE_CYCLES"] / waves)
    entries = []
    for key, t in sorted(traffic.items()):
        if "fetch_kb" in t and "write_kb" in t:
            mlp = key.startswith("mlp_")
            noise = 0.05 if key.endswith("_n5") else 0.0
            spl = 16 if mlp else int(key.split("_")[0][1:])
            envs = int(key.split("_")[1]) if mlp else 65536
            entries.append({"workload": (f"mono {envs} envs, learned policy" if mlp else "mono 65536 envs, random policy" + (", noise 0.05" if noise else "")),
                            "env_kind": "mono", "noise": noise,
                            "policy": "mlp" if mlp else "random", "envs": envs, "kernel": t["kernel"],
                            "steps_per_launch": spl, "FETCH_SIZE_KB": t["fetch_kb"], "WRITE_SIZE_KB": t["write_kb"],
                            # gfx950: FETCH_SIZE tallies 64 B per 128-B request on wide coalesced reads -> doubled
                            "hbm_bytes_per_launch": (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0,
                            "valu_issue_ratio": valu.get(key)})
    out["traffic"] = {"note": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, separate --pmc passes, KB = 1024 B; "
                              "mean per dispatch of the dominant kernel", "entries": entries,
                      "collected_by": "tools/prof_r03.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_*, one counter group per run)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
