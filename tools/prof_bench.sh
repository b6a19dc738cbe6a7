#!/bin/bash
# Profiles `python bench.py` on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
# usage: tools/prof_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-run}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH_ARGS="--steps 256 --warmup 64 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py $BENCH_ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python bench.py $BENCH_ARGS > $OUT/pmc_sq.log 2>&1 || { echo "pmc_sq failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python bench.py $BENCH_ARGS > $OUT/pmc_sq2.log 2>&1 || { echo "pmc_sq2 failed"; tail -5 $OUT/pmc_sq2.log; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py $BENCH_ARGS > $OUT/pmc_fetch.log 2>&1 || { echo "pmc_fetch failed"; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py $BENCH_ARGS > $OUT/pmc_write.log 2>&1 || { echo "pmc_write failed"; tail -5 $OUT/pmc_write.log; }
find $OUT -name "*.csv" | head -40
