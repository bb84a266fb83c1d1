# does running the clean-up of launch group g beside the U-Net of group g+1 pay for the narrow models?  (64 images per step)
for b in 16 32; do
for cfg in "0 " "32 " "32 --overlap" "16 --overlap" "21 --overlap"; do
 set -- $cfg
 python3 bench.py --base $b --images 64 --group $1 $2 --no-cpu-baseline --no-narrow --no-host-inclusive --no-kernel-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base$b group $1 $2', d['value'], d['ms_per_step'], d['stage_ms_per_image'])"
done; done
