#!/bin/bash
for kind in sort press mono; do
 for noise in 0.0 0.05; do
  for pl in 2 0; do
    timeout -k 5 120 python bench.py --policy mlp --kind $kind --noise $noise --envs 65536 --steps 256 --warmup 32 --chunk 16 --pipeline $pl --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kind noise $noise pipeline $pl  %.2f G env-steps/s  launch %.1f us' % (d['value']/1e9, d['roofline']['launch_ms']*1e3))"
  done
 done
done
for n in 8192 32768; do for pl in 2 0; do
    timeout -k 5 120 python bench.py --policy mlp --envs $n --steps 256 --warmup 32 --chunk 16 --pipeline $pl --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mono $n envs pipeline $pl  %.2f G env-steps/s  launch %.1f us' % (d['value']/1e9, d['roofline']['launch_ms']*1e3))"
done; done
