#!/bin/bash
# VALU accounting per role of k_rollout_ring: the same PMC pass over the shipped kernel and over builds with
# one role compiled out (build/abl/libmse_{BASE,RING_NOO,RING_NOQ,NODRAW}.so; never shipped).
OUT=gpurun_out/prof_roles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="--steps 256 --warmup 64 --no-cpu-baseline"
for v in BASE RING_NOO RING_NOQ NODRAW; do
  export MSE_LIB_PATH=$PWD/build/abl/libmse_$v.so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $OUT/$v -- python bench.py $A > $OUT/$v.log 2>&1 || tail -3 $OUT/$v.log
done
unset MSE_LIB_PATH
python - <<PY
import csv,glob,collections
for v in ("BASE","RING_NOO","RING_NOQ","NODRAW"):
    agg=collections.defaultdict(list)
    for p in glob.glob("$OUT/%s/**/*counter_collection.csv"%v, recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_rollout" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, "  ".join("%s=%.0f" % (c, sum(x)/len(x)/1024/16) for c,x in sorted(agg.items())), "(per SIMD per step)")
PY
