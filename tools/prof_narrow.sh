#!/bin/bash
# rocprofv3 kernel statistics of the narrow canonical models (gpurun): bash tools/prof_narrow.sh -> gpurun_out/prof_narrow/kernel_stats_base{16,32}.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_narrow
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 16 32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$b -o b$b -- python3 $R/bench.py --base $b --images $((1024 / b)) --group 0 --no-cpu-baseline --no-host-inclusive --no-narrow --steps 3 --warmup 1 > $O/bench_base$b.json 2> $O/bench_base$b.err
  find $O/t$b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_base$b.csv
  rm -rf $O/t$b
  head -8 $O/kernel_stats_base$b.csv | cut -d, -f1-5 | cut -c1-110
done
