#!/bin/bash
# rocprofv3 kernel statistics of the overlay row (BASELINE configs[4]) and meta_preprocess on 64 resident images (gpurun):
#   bash tools/aux_profile.sh   -> gpurun_out/aux_prof/kernel_stats.csv, legs.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/aux_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o aux -- python3 $R/tools/aux_legs.py --no-comm > $O/legs.json 2> $O/legs.err
cd $R
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace
tail -1 $O/legs.json | cut -c1-300
head -12 $O/kernel_stats.csv | cut -d, -f1-5 | cut -c1-120
