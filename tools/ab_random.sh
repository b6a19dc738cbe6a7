#!/bin/bash
# same-box A/B of two builds of the library on the headline random rollout: build/abl/libmse_old.so against the tree's
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5" "--steps 1024 --warmup 128" "--envs 262144 --steps 512 --warmup 64"; do
  for lib in build/abl/libmse_old.so marl-sortingenv_amd/libmse_hip.so; do
    MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-36s %-40s %.2f G env-steps/s  launch %.1f us' % ('$lib', '$cfg', d['value']/1e9, d['roofline']['launch_ms']*1e3))"
  done
done
done
