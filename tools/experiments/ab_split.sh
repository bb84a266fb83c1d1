for i in 1 2; do
for lib in oldcombine hip; do
  ECSEG_HIP_LIB=$PWD/ecseg_amd/libecseg_$lib.so timeout -k 10 300 python3 bench.py --steps 10 --no-cpu-baseline --no-narrow --no-host-inclusive --wino 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('split_bf16x3',{}).get('value'))"
done; done
