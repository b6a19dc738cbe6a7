#!/bin/bash
# extra PMC passes (one counter group per run)
TAG=${1:-x}; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="--steps 256 --warmup 64 --no-cpu-baseline ${@:2}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_a -- python bench.py $A > $OUT/a.log 2>&1 || tail -3 $OUT/a.log
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc_b -- python bench.py $A > $OUT/b.log 2>&1 || tail -3 $OUT/b.log
rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc_c -- python bench.py $A > $OUT/c.log 2>&1 || tail -3 $OUT/c.log
python - <<PY
import csv,glob,collections
for tag in "abc":
    agg=collections.defaultdict(list)
    for p in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv"%tag, recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_rollout" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in sorted(agg.items()): print("%-32s %16.1f" % (c, sum(v)/len(v)))
PY
