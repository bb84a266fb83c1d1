#!/bin/bash
# rocprofv3 kernel statistics of tools/post_bench.py, one run per input kind (--only speckle / synth): which kernels of
# meta_inference + count take the time on speckled vs realistic label maps -> gpurun_out/pp_<kind>/
R=$PWD
for k in speckle synth; do
mkdir -p $R/gpurun_out/pp_$k
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_$k -o post -- python3 $R/tools/post_bench.py --reps 4 --only $k > $R/gpurun_out/pp_$k/bench.log 2>&1
cd $R
tail -1 gpurun_out/pp_$k/bench.log | cut -c1-400
f=$(find gpurun_out/pp_$k -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'conv' not in r['Name'] and 'Cijk' not in r['Name'] and 'at::' not in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    n=r['Name']
    print('%-44s calls %6s avg %9.1f us  %5.1f%%' % (n.split('(')[0].replace('ecseg::','').replace('void ','')[:44], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
done
