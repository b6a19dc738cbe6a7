#!/bin/bash
# timing ablations (never shipped): each library is the kernel with one part compiled out
for v in BASE NODRAW NOOBS NOPOLICY; do
  for outs in "" "--no-outputs"; do
    MSE_LIB_PATH=$PWD/build/abl/libmse_$v.so timeout -k 5 100 python bench.py --steps 208 --warmup 32 --no-cpu-baseline $outs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$outs', 'us/step(kernel)=%.2f' % (d['roofline']['launch_ms']*1e3/16), 'wall us/step=%.2f' % (d['ms_per_step']*1e3))"
  done
done
