#!/bin/bash
# rocprofv3 kernel statistics of tools/post_bench.py (meta_inference + count on three input kinds) -> gpurun_out/post_prof/
mkdir -p gpurun_out/post_prof
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/post_prof -o post -- python3 $R/tools/post_bench.py --reps 3 > $R/gpurun_out/post_prof/bench.log 2>&1
cd $R
f=$(find gpurun_out/post_prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'conv' not in r['Name'] and 'Cijk' not in r['Name'])
for r in rows[:40]:
    n=r['Name']
    if 'conv' in n: continue
    print('%-44s calls %6s avg %9.1f us  %5.1f%%' % (n.split('(')[0].replace('ecseg::','').replace('void ','')[:44], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
