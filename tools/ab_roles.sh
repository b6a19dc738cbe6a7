#!/bin/bash
# same-box A/B of the learned-policy rollout kernel shapes in ONE build: plain (--pipeline 2), actor + critic waves (1),
# actor + critic + RNG waves (0 = by size), alternating
for rep in 1 2 3; do
for n in ${SIZES:-65536}; do
  for pl in 2 1 0; do
    timeout -k 5 120 python bench.py --policy mlp --envs $n --steps 256 --warmup 32 --chunk ${CHUNK:-16} --pipeline $pl --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipeline $pl  %7d envs  %.2f G env-steps/s  launch %.1f us  %s' % ($n, d['value']/1e9, d['roofline']['launch_ms']*1e3, d['roofline']['kernel']))"
  done
done
done
