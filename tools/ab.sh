#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh OLD_LIB "bench args" ["bench args" ...]
# (alternating runs, two rounds; the tree's libmse_hip.so is the candidate)
OLD=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  for lib in $OLD marl-sortingenv_amd/libmse_hip.so; do
    MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py $cfg --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-36s %-44s %.2f G env-steps/s  launch %.1f us' % ('$lib', '$cfg', d['value']/1e9, d['roofline']['launch_ms']*1e3))"
  done
done
done
