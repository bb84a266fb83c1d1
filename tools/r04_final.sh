#!/bin/bash
# Round-4 measurement set (GPU box): bench line, layer tables of the three models, post-processing by input kind + its kernel
# statistics and FETCH / WRITE traffic, CLI rates, host scaling.  -> gpurun_out/r04_final/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py --layer-table $O/layer_table.json > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for b in 32 16; do
  timeout -k 10 200 python3 bench.py --base $b --images $((1024 / b)) --group 0 --no-cpu-baseline --no-narrow --no-host-inclusive --layer-table $O/layer_table_base$b.json > $O/bench_base$b.json 2> $O/bench_base$b.err; echo "bench base $b rc $?"
done
timeout -k 10 200 python3 tools/post_bench.py --reps 5 > $O/post_bench.json 2> $O/post_bench.err; echo "post_bench rc $?"
for b in 16 64; do
  timeout -k 10 200 python3 tools/time_cli.py --base $b --batch $((b == 16 ? 32 : 16)) >> $O/cli_timing.jsonl 2>> $O/cli_timing.err; echo "cli base $b rc $?"
done
timeout -k 10 300 python3 tools/host_scaling.py --n 1024 --ranks 4 --work /tmp/ecseg_hs > $O/host_scaling_gpubox.json 2> $O/host_scaling.err; echo "host scaling rc $?"
