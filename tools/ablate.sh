#!/bin/bash
# ablation TIMING of the three-role rollout kernel (results of the ablated builds are invalid by construction):
# how the step time responds when one role loses a chunk of its work
for lib in marl-sortingenv_amd/libmse_hip.so build/abl/libmse_abl_NOPRESS.so build/abl/libmse_abl_NORNG.so build/abl/libmse_abl_NOOBS.so; do
  MSE_LIB_PATH=$PWD/$lib timeout -k 5 120 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.2f G env-steps/s  launch %.1f us  %.3f us/step' % ('$lib', d['value']/1e9, d['roofline']['launch_ms']*1e3, d['ms_per_step']*1e3))"
done
