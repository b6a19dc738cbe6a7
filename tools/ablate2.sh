#!/bin/bash
run() { lib=$1; shift; MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline --no-configs "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %-24s %.2f G env-steps/s  launch %.1f us  %.3f us/step' % ('$lib', '$*', d['value']/1e9, d['roofline']['launch_ms']*1e3, d['ms_per_step']*1e3))"; }
for lib in marl-sortingenv_amd/libmse_hip.so build/abl/libmse_abl_NOPRESS.so build/abl/libmse_abl_NOOBS.so; do
  run $lib
  run $lib --no-outputs
done
run marl-sortingenv_amd/libmse_hip.so --pipeline 1
run marl-sortingenv_amd/libmse_hip.so --pipeline 2
run marl-sortingenv_amd/libmse_hip.so --pipeline 2 --no-outputs
run marl-sortingenv_amd/libmse_hip.so --envs 1048576 --steps 256 --warmup 64
run marl-sortingenv_amd/libmse_hip.so --envs 1048576 --steps 256 --warmup 64 --no-outputs
