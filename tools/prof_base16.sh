mkdir -p gpurun_out/p16; R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p16 -o b16 -- python3 $R/bench.py --base 16 --no-cpu-baseline --no-host-inclusive --steps 4 --warmup 1 > $R/gpurun_out/p16/log.txt 2>&1
cd $R; tail -1 gpurun_out/p16/log.txt | cut -c1-120
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/p16/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-60s calls %5s avg %9.1f us %5.1f%%' % (r['Name'].split('(')[0].replace('ecseg::','').replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print('total ms per step', tot/5/1e6)
PY
