#!/bin/bash
# launch time of the learned-policy rollout against steps per launch: T(K) = edges + K x step  [ENVS=65536] [PIPELINE=0]
for k in 1 2 4 8 16 32 64; do
  timeout -k 5 120 python bench.py --policy mlp --envs ${ENVS:-65536} --steps $((16*k)) --warmup $k --chunk $k --pipeline ${PIPELINE:-0} --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=%-3d launch %.2f us  (%.2f G)  %s' % ($k, d['roofline']['launch_ms']*1e3, d['value']/1e9, d['roofline']['kernel']))"
done
