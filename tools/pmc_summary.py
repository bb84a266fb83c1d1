#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output) into per-kernel HBM traffic.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [base=64,up=transpose,wino=2]

The summary records the configuration of the profiled command and the sha256 of the kernel sources it ran
(ecseg_amd.build.source_hash): bench.py quotes `roofline.traffic` only from a summary of the same configuration AND the
same sources, and prints `traffic: null, traffic_stale: true` otherwise.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d):
    rows = list(csv.DictReader(open(glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0])))
    agg = collections.OrderedDict()
    for r in rows:
        a = agg.setdefault(r['Kernel_Name'], [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch), load(write)
    from ecseg_amd.build import source_hash
    cfg = dict(kv.split('=') for kv in sys.argv[4].split(',')) if len(sys.argv) > 4 else {'base': '64', 'up': 'transpose', 'wino': '2'}
    res = {'config': cfg, 'kernel_source_sha256': source_hash(), 'units': 'bytes', 'fetch_correction': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads)',
           'write_correction': 'WRITE_SIZE KiB x 1024', 'kernels': {}}
    tot_conv = [0, 0.0, 0.0]
    for k in f:
        n = f[k][0]
        fb = f[k][1] * 1024 * 2
        wb = w.get(k, [n, 0.0])[1] * 1024
        res['kernels'][k] = {'launches': n, 'fetch_bytes_per_launch': fb / n, 'write_bytes_per_launch': wb / n,
                             'hbm_bytes_per_launch': (fb + wb) / n}
        if 'conv_mfma_kernel' in k or 'conv_wino' in k or 'convs_kernel' in k:
            tot_conv[0] += n; tot_conv[1] += fb; tot_conv[2] += wb
    if tot_conv[0]:
        res['conv_mfma_all'] = {'launches': tot_conv[0], 'hbm_bytes_per_launch': (tot_conv[1] + tot_conv[2]) / tot_conv[0],
                                'fetch_bytes_per_launch': tot_conv[1] / tot_conv[0],
                                'write_bytes_per_launch': tot_conv[2] / tot_conv[0]}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res.get('conv_mfma_all')))


if __name__ == '__main__':
    main()
