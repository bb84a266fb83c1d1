#!/bin/bash
# A/B harness for `make metaseg` host settings on ONE GPU box (round 5): every variant runs tools/time_cli.py over the same
# generated inputs (--keep reuses them; outputs of the run before are removed: overwriting files is slower than creating them),
# outputs on tmpfs (the boxes' disk-backed /tmp throttles dirty pages after a few GB and then decides the result).
#   gpurun -- 'bash tools/experiments/cli_ab.sh /dev/shm/ecseg_keep 1024'
# What it found: cpu.max = "1600000 100000" - a 16-CPU quota; the base-16 loop is CPU-bound (cpus_busy ~15.5, throttled periods),
# ECSEG_DEBUG_CALLS=1 shows the device calls stretched by the throttled periods, not by the GPU.
set -e
K=${1:-/dev/shm/ecseg_keep}
N=${2:-1024}
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   cpus visible: $(nproc)"
run() { tag=$1; shift; python3 tools/time_cli.py --keep $K --n $N --base 16 --batch 32 "$@" > gpurun_out/cli_ab_$tag.json 2> gpurun_out/cli_ab_$tag.err; python3 - <<PY
import json
d = json.load(open('gpurun_out/cli_ab_$tag.json'))
print('$tag', d['images_per_s'], 'device calls', d['device_call_seconds'], 's', d['cgroup_cpu'], d['host_stage_ms_per_image'], d['pinned_pool'])
PY
}
run default
run pageable --pinned-mb 0
run io16 --io-threads 16
run workers2 --workers 2
run smooth --weights smooth
ECSEG_SPIN_WAIT=1 run spin
rm -rf $K
