for args in "--base 16 --images 64 --group 0" "--base 16 --images 64 --group 0 --overlap" "--base 16 --images 64 --group 16 --overlap" "--base 16 --images 64 --group 8 --overlap" "--base 32 --images 32 --group 0" "--base 32 --images 32 --group 0 --overlap" "--base 32 --images 32 --group 8 --overlap" "--base 64 --images 16" "--base 64 --images 16 --group 8 --overlap" "--base 64 --images 16 --group 8"; do
  timeout -k 10 300 python3 bench.py $args --steps 8 --no-cpu-baseline --no-narrow --no-host-inclusive --no-kernel-profile 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', '->', d['value'], d['ms_per_step'], d['stage_ms_per_image'])"
done
