#!/usr/bin/env python3
"""Timeline of ONE pipeline call from a rocprofv3 --kernel-trace CSV: every kernel from the last `tile_patches` launch on,
with start offset, duration and the gap to the previous kernel's end (any stream) - where a single-image call spends its
time between kernels.   python tools/trace_timeline.py <dir with *_kernel_trace.csv> [--post-only]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    post_only = '--post-only' in sys.argv
    f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows))
    last = max(i for i, k in enumerate(ks) if 'tile_patches' in k[2])
    ks = ks[last:]
    t0 = ks[0][0]
    end_prev = t0
    busy = 0
    tot_gap = 0
    agg = {}
    started_post = False
    for s, e, name, q in ks:
        short = name.split('(')[0].replace('ecseg::', '').replace('void ', '')[:60]
        if 'stitch_argmax' in name:
            started_post = True
        gap = s - end_prev
        if not post_only or started_post:
            print('%9.1f us  dur %8.1f  gap %7.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, short))
            a = agg.setdefault(short, [0, 0.0])
            a[0] += 1; a[1] += (e - s) / 1e3
            busy += e - s
            tot_gap += max(gap, 0)
        end_prev = max(end_prev, e)
    print('total %.1f us, kernels busy %.1f us (overlap counted twice), gaps %.1f us' % ((end_prev - t0) / 1e3, busy / 1e3, tot_gap / 1e3))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('  %4d x %-60s %9.1f us' % (n, k, t))


if __name__ == '__main__':
    main()
