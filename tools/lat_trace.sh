#!/bin/bash
# Kernel timeline of a single-image call (gpurun): rocprofv3 kernel trace of tools/latency_probe.py, then tools/trace_timeline.py
#   bash tools/lat_trace.sh <tag>      -> gpurun_out/lat_<tag>/timeline.txt, timeline_post.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lat_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o lat -- python3 $R/tools/latency_probe.py --lanes 2 --images 1 > $OUT/probe.json 2> $OUT/probe.err
cd $R
python3 tools/trace_timeline.py $OUT/trace --post-only > $OUT/timeline_post.txt
python3 tools/trace_timeline.py $OUT/trace > $OUT/timeline.txt
rm -rf $OUT/trace
grep -A40 "^total" $OUT/timeline_post.txt
