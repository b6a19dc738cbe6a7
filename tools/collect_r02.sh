#!/bin/bash
# copies the summaries of a tools/prof_r02.sh run (gpurun_out/prof_r02, scratch) into profiles/r02 (tracked)
set -e
SRC=gpurun_out/prof_r02; DST=profiles/r02; mkdir -p $DST
for t in trace_k64 trace_k20 trace_k16 trace_mlp_65536 trace_mlp_262144; do
  # newest file: gpurun merges every run's output directory into gpurun_out/
  f=$(find $SRC/$t -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2)
  cp "$f" $DST/${t}_kernel_stats.csv; cp $SRC/$t.json $DST/${t}_bench.json
done
cp $SRC/summary.json $DST/summary.json; cp $SRC/configs.txt $DST/configs.txt
for f in free_driver_line free_k64 free_mlp_65536 free_mlp_262144 free_mlp_262144_k64; do cp $SRC/$f.json $DST/$f.json; done
python3 - <<'PY'
import json
s = json.load(open("profiles/r02/summary.json"))
t = s["traffic"]
t["collected_by"] = "tools/prof_r02.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, one counter per run)"
json.dump(t, open("profiles/r02/traffic.json", "w"), indent=1)
for tag, e in s["tags"].items():
    ks = e.get("kernel_stats") or []
    line = tag.ljust(20)
    for k in ks[:1]:
        line += " calls=%s avg=%.1fus min=%.1f max=%.1f" % (k["Calls"], float(k["AverageNs"]) / 1e3, float(k["MinNs"]) / 1e3, float(k["MaxNs"]) / 1e3)
    b = e.get("bench")
    if b:
        line += " | bench %.2f G launch_ms %.4f spl %s frac %.3f" % (b["value"] / 1e9, b["launch_ms"], b["steps_per_launch"], b["frac"])
    if tag.startswith("trace"):
        print(line)
sq = s["tags"]["sq_k64"]["counters_mean_per_dispatch"]["void k_rollout_ring<3, false>"]
print("ring: VALU issue ratio %.3f, VALU per SIMD-step %.0f" % (sq["SQ_ACTIVE_INST_VALU"] / (sq["SQ_WAVE_CYCLES"] / 3), sq["SQ_INSTS_VALU"] / 64 / 1024))
m = [v for k, v in s["tags"]["sq_mlp_262144"]["counters_mean_per_dispatch"].items() if "policy" in k][0]
print("policy: VALU issue ratio %.3f, VALU per wave-step %.0f, MFMA per launch %.0f" % (m["SQ_ACTIVE_INST_VALU"] / (m["SQ_WAVE_CYCLES"] / 2), m["SQ_INSTS_VALU"] / 4096 / 16, m["SQ_INSTS_MFMA"]))
for e in t["entries"]:
    print(e["policy"], e["envs"], "K", e["steps_per_launch"], "MB %.1f" % (e["hbm_bytes_per_launch"] / 1e6), "B/env-step %.1f" % (e["hbm_bytes_per_launch"] / e["envs"] / e["steps_per_launch"]))
PY
for f in free_driver_line free_k64 free_mlp_65536 free_mlp_262144 free_mlp_262144_k64; do python3 -c "
import json; d=json.loads([l for l in open('profiles/r02/$f.json') if l.startswith('{')][-1]); r=d['roofline']; print('$f'.ljust(24), '%.2f G'%(d['value']/1e9), 'launch_ms %.4f'%r['launch_ms'], 'frac %.3f'%r['frac'], 'frac_contract %.3f'%r['frac_contract'], 'traffic', r['traffic'])"; done
cat $DST/configs.txt
