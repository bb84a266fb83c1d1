#!/bin/bash
# A/B/C... timing of library builds on one box: tools/ab.sh <rounds> <lib1> <lib2> ...   (lib = file under ecseg_amd/)
R=$1; shift
for i in $(seq $R); do
  for l in "$@"; do
    v=$(ECSEG_HIP_LIB=$GRAFT_REPO_ROOT/ecseg_amd/$l timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "$l $v"
  done
done
