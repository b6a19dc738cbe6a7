#!/bin/bash
set -o pipefail
OUT=gpurun_out/prof_roles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="--no-cpu-baseline --no-configs"
for pl in 0 2; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq_p$pl -- python3 bench.py $B --policy mlp --envs 65536 --chunk 16 --steps 128 --warmup 16 --reps 1 --pipeline $pl > $OUT/sq_p$pl.json 2> $OUT/sq_p$pl.err || { echo failed; tail -3 $OUT/sq_p$pl.err; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/sq_p$pl/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k,v in acc.items():
    if "rollout_policy" in k:
        d=n[k]; print("pipeline $pl",k,d,"dispatches; per dispatch:",{c:round(x/d) for c,x in v.items()})
PY
done
