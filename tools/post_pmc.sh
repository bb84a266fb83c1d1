#!/bin/bash
# rocprofv3 PMC passes over tools/post_bench.py (instruction mix + waits of the post-processing kernels) -> gpurun_out/post_pmc/
mkdir -p gpurun_out/post_pmc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/post_pmc/mix -o post -- python3 $R/tools/post_bench.py --reps 1 --images 16 > $R/gpurun_out/post_pmc/mix.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/post_pmc/wait -o post -- python3 $R/tools/post_bench.py --reps 1 --images 16 > $R/gpurun_out/post_pmc/wait.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
for sub in ('mix','wait'):
    f=glob.glob('gpurun_out/post_pmc/%s/**/*counter_collection.csv'%sub, recursive=True)
    if not f: print(sub,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0].replace('ecseg::','').replace('void ','')
        if 'ccl' not in k and 'apply' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        print(sub, '%-28s'%k[:28], ' '.join('%s=%.3g'%(c.replace('SQ_',''),x) for c,x in sorted(v.items())))
PY
