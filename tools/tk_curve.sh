#!/bin/bash
# launch time T(K) of the rollout kernel against the steps per launch K, for one or two builds of the library
for K in 1 2 3 4 6 8 12 16 20 32 64; do
  for lib in "$@"; do
    MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py --chunk $K --steps $K --warmup $K --reps 3000 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-36s K=%-3d launch %.2f us  (%.2f G)' % ('$lib', $K, d['roofline']['launch_ms']*1e3, d['value']/1e9))"
  done
done
