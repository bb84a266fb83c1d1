#!/bin/bash
# A/B timing of (library build, option set) pairs on one box, with per-layer tables:
#   tools/ab_cfg.sh <outdir> <rounds> "<name>|<lib under ecseg_amd/>|<bench.py args>" ...
# Each run writes <outdir>/<name>_<round>.json (the bench line) and <outdir>/lt_<name>_<round>.json (per-layer table).
OUT=$1; R=$2; shift 2
mkdir -p $OUT
for i in $(seq $R); do
  for c in "$@"; do
    IFS='|' read -r name lib args <<< "$c"
    ECSEG_HIP_LIB=$GRAFT_REPO_ROOT/ecseg_amd/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-narrow --no-host-inclusive \
        --layer-table $OUT/lt_${name}_$i.json $args > $OUT/${name}_$i.json 2> $OUT/${name}_$i.err || { echo "$name round $i FAILED"; tail -3 $OUT/${name}_$i.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('$OUT/${name}_$i.json').read().strip().splitlines()[-1]); print('$name', $i, d['value'], d['roofline']['frac'])"
  done
done
