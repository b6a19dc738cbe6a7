for v in BASE waves3 waves4; do
  if [ $v = BASE ]; then unset MSE_LIB_PATH; else export MSE_LIB_PATH=$PWD/build/abl/libmse_$v.so; fi
  for n in 262144 1048576; do
    python bench.py --no-cpu-baseline --envs $n --steps 64 --warmup 16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v envs $n: G steps/s %6.2f launch_ms %.4f frac %.3f %s' % (d['value']/1e9, r['launch_ms'], r['frac'], r['kernel']))"
  done
done
