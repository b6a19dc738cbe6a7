#!/bin/bash
# Round-3 profiles on the GPU box (one gpurun call): kernel traces + PMC passes of bench.py at the steps-per-launch values
# the roofline quotes (20 = the driver's line, 64), the class-default noise, the learned-policy rollout.  Summaries ->
# gpurun_out/prof_r03/ (copy into profiles/r03/).  Counters are collected in their own runs, one group per run, with
# --kernel-trace only (never with --sys-trace / runtime traces).
set -o pipefail
OUT=gpurun_out/prof_r03; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="--no-cpu-baseline --no-configs"
run_trace() { # tag, bench args
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 bench.py $B "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag failed"; tail -3 $OUT/$tag.err; }
}
run_pmc() { # tag, counters (space separated), bench args
  local tag=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$tag -- python3 bench.py $B "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag failed"; tail -3 $OUT/$tag.err; }
}
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
run_trace trace_k20 --steps 20 --warmup 5
run_trace trace_k64 --steps 1024 --warmup 128
run_trace trace_k20_n5 --steps 20 --warmup 5 --noise 0.05
for K in 20 64; do
  run_pmc fetch_k$K FETCH_SIZE --chunk $K --steps $((K*8)) --warmup $K --reps 1
  run_pmc write_k$K WRITE_SIZE --chunk $K --steps $((K*8)) --warmup $K --reps 1
  run_pmc sq_k$K "$SQ" --chunk $K --steps $((K*8)) --warmup $K --reps 1
done
run_pmc fetch_k20_n5 FETCH_SIZE --chunk 20 --steps 160 --warmup 20 --reps 1 --noise 0.05
run_pmc write_k20_n5 WRITE_SIZE --chunk 20 --steps 160 --warmup 20 --reps 1 --noise 0.05
run_pmc sq_k20_n5 "$SQ" --chunk 20 --steps 160 --warmup 20 --reps 1 --noise 0.05
# learned-policy rollout (configs[3]-shaped collection), both sizes
run_trace trace_mlp_65536 --policy mlp --envs 65536 --chunk 16 --steps 256 --warmup 32
run_trace trace_mlp_262144 --policy mlp --envs 262144 --chunk 16 --steps 256 --warmup 32
run_pmc sq_mlp_262144 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" --policy mlp --envs 262144 --chunk 16 --steps 128 --warmup 16 --reps 1
run_pmc sq_mlp_65536 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" --policy mlp --envs 65536 --chunk 16 --steps 128 --warmup 16 --reps 1
run_pmc fetch_mlp_262144 FETCH_SIZE --policy mlp --envs 262144 --chunk 16 --steps 128 --warmup 16 --reps 1
run_pmc write_mlp_262144 WRITE_SIZE --policy mlp --envs 262144 --chunk 16 --steps 128 --warmup 16 --reps 1
# free-running reference points (no profiler attached): the driver's line with its sub-runs, and a 64-step collector
python3 bench.py --steps 20 --warmup 5 > $OUT/free_driver_line.json 2> $OUT/free.err
python3 bench.py $B > $OUT/free_k64.json 2>> $OUT/free.err
python3 tools/summarize_r03.py $OUT > $OUT/summary.json
# keep the merge small: drop the raw per-dispatch traces, keep stats and counter CSVs
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
tail -c 1200 $OUT/summary.json
