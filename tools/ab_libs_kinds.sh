#!/bin/bash
# same-box A/B of library builds on the learned-policy rollout over env kinds and noise: LIBS="a.so b.so"
for kind in sort press mono; do
 for noise in 0.0 0.05; do
  for lib in $LIBS; do
    MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py --policy mlp --kind $kind --noise $noise --envs 65536 --steps 256 --warmup 32 --chunk 16 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-36s $kind noise $noise  %.2f G env-steps/s  launch %.1f us' % ('$lib', d['value']/1e9, d['roofline']['launch_ms']*1e3))"
  done
 done
done
