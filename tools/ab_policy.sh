#!/bin/bash
# same-box A/B of two builds of the library on the learned-policy rollout: build/abl/libmse_old.so against the tree's
for rep in 1 2; do
for n in 65536 262144; do
  for lib in build/abl/libmse_old.so marl-sortingenv_amd/libmse_hip.so; do
    MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py --policy mlp --envs $n --steps 256 --warmup 32 --chunk 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-36s %7d envs  %.2f G env-steps/s  launch %.1f us' % ('$lib', $n, d['value']/1e9, d['roofline']['launch_ms']*1e3))"
  done
done
done
