#!/bin/bash
one() { # lib chunk envs
  MSE_LIB_PATH=$PWD/$1 timeout -k 5 120 python bench.py --policy mlp --envs $3 --steps $((16*$2)) --warmup $2 --chunk $2 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s K=%-3d %7d envs  %.2f G  launch %.1f us' % ('$1', $2, $3, d['value']/1e9, d['roofline']['launch_ms']*1e3))"
}
for rep in 1 2; do
for lib in build/abl/libmse_prio0.so marl-sortingenv_amd/libmse_hip.so build/abl/libmse_prio3.so; do one $lib 16 65536; done
done
for k in 4 8 32 64; do one marl-sortingenv_amd/libmse_hip.so $k 65536; done
for n in 16384 32768 49152; do one marl-sortingenv_amd/libmse_hip.so 16 $n; done
