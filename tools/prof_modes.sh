#!/bin/bash
# kernel-trace of the per-step launch modes (rollout K=1 and sample+step)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k1 -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --chunk 1 > gpurun_out/prof_k1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --mode step > gpurun_out/prof_step.log 2>&1
python - <<PY
import csv,glob
for d in ("prof_k1","prof_step"):
    for p in glob.glob("gpurun_out/%s/**/*kernel_stats.csv"%d, recursive=True):
        for r in list(csv.DictReader(open(p)))[:4]: print(d, r["Name"][:50], r["Calls"], r["AverageNs"])
PY
tail -1 gpurun_out/prof_step.log | cut -c1-160
