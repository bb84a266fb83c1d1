#!/bin/bash
# Round profile set (run on the GPU box through gpurun): kernel trace + stats, then PMC passes in their own runs
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never together with --sys-trace etc.).
#   bash tools/profile_round.sh <tag> [bench args, e.g. --base 16]     -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r02}
shift
XARGS="$*"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py $XARGS --no-cpu-baseline --no-host-inclusive --steps 4 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
echo "stats rc $?"
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt
PMC1="python3 $R/bench.py $XARGS --no-cpu-baseline --no-host-inclusive --steps 1 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- $PMC1 > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc $?"
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o bench -- $PMC1 > $OUT/pmc_mfma.log 2>&1
echo "pmc mfma rc $?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_mix -o bench -- $PMC1 > $OUT/pmc_mix.log 2>&1
echo "pmc mix rc $?"
cd $R
# configuration key of the summary (bench.py matches it): base / up / wino from the bench arguments
CFG=$(python3 - $XARGS <<'PY'
import sys
a = sys.argv[1:]
def opt(name, d):
    return a[a.index(name) + 1] if name in a else d
print('base=%s,up=%s,wino=%s' % (opt('--base', '64'), opt('--up', 'transpose'), '0' if '--direct' in a else opt('--wino', '2')))
PY
)
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_traffic.json $CFG
python3 tools/pmc_mfma_summary.py $OUT $OUT/mfma_busy.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
ls $OUT
