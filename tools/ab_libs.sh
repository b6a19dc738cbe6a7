#!/bin/bash
# same-box A/B of library builds on the learned-policy rollout: LIBS="a.so b.so" [SIZES="65536"] [CHUNK=16] [PIPELINE=0]
for rep in 1 2 3; do
for n in ${SIZES:-65536}; do
  for lib in $LIBS; do
    MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py --policy mlp --envs $n --steps $((16*${CHUNK:-16})) --warmup ${CHUNK:-16} --chunk ${CHUNK:-16} --pipeline ${PIPELINE:-0} --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %7d envs  %.2f G env-steps/s  launch %.1f us' % ('$lib', $n, d['value']/1e9, d['roofline']['launch_ms']*1e3))"
  done
done
done
