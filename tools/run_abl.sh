#!/bin/bash
# timing ablations (never shipped): each library is the kernel with one part compiled out
for f in build/abl/libmse_*.so; do
  v=$(basename $f .so); v=${v#libmse_}
  for outs in "" "--no-outputs"; do
    MSE_LIB_PATH=$PWD/$f timeout -k 5 100 python bench.py --steps 208 --warmup 32 --no-cpu-baseline $outs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-13s kernel us/step=%.2f' % ('$v', '$outs', d['roofline']['launch_ms']*1e3/16))"
  done
done
