#!/bin/bash
# Per-layer HBM traffic of the default build and of block-order variants: for every library given, one
# FETCH_SIZE and one WRITE_SIZE pass of a 1-step bench + a layer table, matched by tools/pmc_layer_traffic.py.
#   bash tools/traffic_variants.sh <outdir> <lib under ecseg_amd/> ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
for lib in "$@"; do
  tag=$(echo $lib | tr -c 'A-Za-z0-9\n' '_')
  export ECSEG_HIP_LIB=$R/ecseg_amd/$lib
  cd $R && timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-host-inclusive --no-narrow --steps 4 --warmup 1 --layer-table $OUT/lt_$tag.json > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${c}_$tag -o bench -- python3 $R/bench.py --no-cpu-baseline --no-host-inclusive --no-narrow --steps 1 --warmup 1 > $OUT/pmc_${c}_$tag.log 2>&1
    echo "$tag $c rc $?"
  done
  cd $R
  f=$(find $OUT/pmc_FETCH_SIZE_$tag -name '*counter_collection.csv' | head -1)
  w=$(find $OUT/pmc_WRITE_SIZE_$tag -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_layer_traffic.py $f $w $OUT/lt_$tag.json $OUT/layer_traffic_$tag.json | tail -3
done
