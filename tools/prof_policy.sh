#!/bin/bash
# kernel trace of tools/bench_policy.py (k_policy_mlp average duration)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_policy -- python tools/bench_policy.py --iters 50 > gpurun_out/prof_policy.log 2>&1
python - <<PY
import csv,glob
for p in glob.glob("gpurun_out/prof_policy/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:3]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
tail -1 gpurun_out/prof_policy.log | cut -c1-400
