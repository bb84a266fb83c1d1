for l in 1 2; do
 python3 bench.py --opt unet_lanes=$l --no-cpu-baseline --no-narrow --no-host-inclusive --no-kernel-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base64 lanes $l', d['value'], d['stage_ms_per_image'])"
 for b in 32 16; do
 python3 bench.py --base $b --images $((1024 / b)) --group 0 --opt unet_lanes=$l --no-cpu-baseline --no-narrow --no-host-inclusive --no-kernel-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base$b lanes $l', d['value'], d['stage_ms_per_image'])"
 done
done
