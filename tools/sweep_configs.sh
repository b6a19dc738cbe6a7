#!/bin/bash
# bench.py over the parity-test configurations (not bench lines: context for DESIGN.md's tables)
for args in "--kind mono --envs 65536" "--kind sort --envs 65536" "--kind press --envs 65536" "--kind mono --envs 131072" \
            "--kind mono --envs 262144" "--kind mono --envs 1048576" "--kind mono --envs 65536 --noise 0.05" \
            "--kind mono --envs 65536 --chunk 1" "--kind mono --envs 65536 --mode step"; do
  python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
f=lambda v, fmt: '  n/a ' if v is None else fmt % v
print('%-42s | G steps/s %6.2f | ms/step %.4f | launch_ms %s | frac %s | %s' % ('$args', d['value']/1e9, d['ms_per_step'], f(r['launch_ms'], '%.4f'), f(r['frac'], '%.3f'), r['kernel']))"
done
