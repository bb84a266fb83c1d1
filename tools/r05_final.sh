#!/bin/bash
# Round-5 measurement set (GPU box, `gpurun -- bash tools/r05_final.sh <part>`): part a = the bench line and layer tables of the three
# models, clean-up by input kind with its kernel statistics, single-image timeline, CLI rates; part b = rocprofv3 kernel
# statistics + PMC passes (FETCH / WRITE traffic, MFMA busy, instruction mix) of the default bench command, kernel statistics of
# the narrow models.  -> gpurun_out/r05_final/ (copied into profiles/r05_* by hand).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
if [ "${1:-a}" = a ]; then
  timeout -k 10 400 python3 bench.py --layer-table $O/layer_table.json > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
  for b in 32 16; do
    timeout -k 10 200 python3 bench.py --base $b --images $((1024 / b)) --group 0 --no-cpu-baseline --no-narrow --no-host-inclusive --layer-table $O/layer_table_base$b.json > $O/bench_base$b.json 2> $O/bench_base$b.err; echo "bench base $b rc $?"
  done
  timeout -k 10 300 python3 tools/post_bench.py --reps 5 > $O/post_bench.json 2> $O/post_bench.err; echo "post_bench rc $?"
  timeout -k 10 600 bash tools/post_profile_kinds.sh > $O/post_kernels_by_kind.txt 2>&1; echo "post kinds rc $?"
  for k in speckle synth; do f=$(find gpurun_out/pp_$k -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/post_kernel_stats_$k.csv; done
  bash tools/lat_trace.sh r05 > $O/lat_trace.log 2>&1; cp gpurun_out/lat_r05/timeline_post.txt $O/single_image_timeline_post.txt; cp gpurun_out/lat_r05/timeline.txt $O/single_image_timeline.txt; cp gpurun_out/lat_r05/probe.json $O/latency_probe.json
  for b in 16 64; do
    timeout -k 10 300 python3 tools/time_cli.py --base $b --batch $((b == 16 ? 32 : 16)) >> $O/cli_timing.jsonl 2>> $O/cli_timing.err; echo "cli base $b rc $?"
  done
  # the narrow model again on a job long enough to leave the start-up regime (1024 files; outputs on tmpfs: 15 GB)
  timeout -k 10 400 python3 tools/time_cli.py --keep /dev/shm/ecseg_cli --n 1024 --base 16 --batch 32 >> $O/cli_timing.jsonl 2>> $O/cli_timing.err; echo "cli base 16 x1024 rc $?"
  rm -rf /dev/shm/ecseg_cli
  # ... and with the fitted smooth-output stand-in for a trained model (blobs instead of speckle: what the label PNG coder sees in real use)
  for n in 256 1024; do timeout -k 10 400 python3 tools/time_cli.py --keep /dev/shm/ecseg_cli --n $n --base 16 --batch 32 --weights smooth >> $O/cli_timing.jsonl 2>> $O/cli_timing.err; echo "cli smooth x$n rc $?"; done
  rm -rf /dev/shm/ecseg_cli
  # ... and with uncompressed input TIFFs (no LZW decode: the loop is then bound by the device, not by the CPU quota)
  for w in random smooth; do timeout -k 10 400 python3 tools/time_cli.py --keep /dev/shm/ecseg_cli --n 1024 --base 16 --batch 32 --input-compression raw --weights $w >> $O/cli_timing.jsonl 2>> $O/cli_timing.err; echo "cli raw $w rc $?"; done
  rm -rf /dev/shm/ecseg_cli
  timeout -k 10 100 python3 tools/experiments/host_call_probe.py --base 16 --batch 32 > $O/host_call_probe.json 2>> $O/cli_timing.err; echo "host probe rc $?"
  timeout -k 10 100 python3 tools/experiments/wait_cpu.py 1 > $O/wait_cpu.jsonl 2>> $O/cli_timing.err; ECSEG_SPIN_WAIT=1 timeout -k 10 100 python3 tools/experiments/wait_cpu.py 0 >> $O/wait_cpu.jsonl 2>> $O/cli_timing.err
else
  bash tools/profile_round.sh r05 --no-narrow > $O/profile_round.log 2>&1; echo "profile_round rc $?"
  for f in kernel_stats.csv pmc_traffic.json mfma_busy.json; do cp gpurun_out/prof_r05/$f $O/ 2>/dev/null; done
  for c in FETCH_SIZE WRITE_SIZE; do f=$(find gpurun_out/prof_r05/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 - "$f" $O/pmc_$(echo $c | tr A-Z a-z).csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    f.write('kernel,dispatches,counter_sum\n')
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%.0f\n' % (k, n, v))
PY
  done
  bash tools/prof_narrow.sh > $O/prof_narrow.log 2>&1; cp gpurun_out/prof_narrow/kernel_stats_base16.csv $O/kernel_stats_base16.csv; cp gpurun_out/prof_narrow/kernel_stats_base32.csv $O/kernel_stats_base32.csv
fi
ls $O
