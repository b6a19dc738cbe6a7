#!/bin/bash
# same-box A/B of alternative builds of the library (MSE_LIB_PATH=build/abl/libmse_<x>.so): the way two builds are compared
# (DESIGN.md section 6); the round-1 ablation switches themselves are gone from the kernels (results: profiles/r01/)
for f in build/abl/libmse_*.so; do
  v=$(basename $f .so); v=${v#libmse_}
  for outs in "" "--no-outputs"; do
    MSE_LIB_PATH=$PWD/$f timeout -k 5 100 python bench.py --steps 208 --warmup 32 --no-cpu-baseline $outs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-13s kernel us/step=%.2f' % ('$v', '$outs', d['roofline']['launch_ms']*1e3/d['config']['steps_per_launch']))"
  done
done
