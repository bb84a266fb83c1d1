#!/bin/bash
# Kernel timeline of the clean-up of a 64-image launch group (base 16): are there gaps between its 32 kernels?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/batch_post
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o lat -- python3 $R/tools/latency_probe.py --base 16 --lanes 1 --images 64 > $OUT/probe.json 2> $OUT/probe.err
cd $R
python3 tools/trace_timeline.py $OUT/trace --post-only > $OUT/timeline_post.txt
rm -rf $OUT/trace
cat $OUT/timeline_post.txt | head -60
