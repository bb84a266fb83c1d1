#!/usr/bin/env python3
"""Per-layer HBM traffic of the MFMA convolutions: the per-dispatch FETCH_SIZE / WRITE_SIZE rows of the two rocprofv3
--pmc passes of tools/profile_round.sh, matched to the layers of profiles/<tag>_layer_table.json by dispatch order (the
MFMA-convolution launches of one segment call are the plan's convolutions in order), against the layer's algorithmic
bytes (input + output tensors once + filter, fp32; cropped layers: the fraction of regions computed).

    python tools/pmc_layer_traffic.py profiles/r02_pmc_fetch_size.csv profiles/r02_pmc_write_size.csv \
           profiles/r02_layer_table.json profiles/r02_layer_traffic.json
"""
import csv
import json
import sys


def conv_rows(path):
    rows = [r for r in csv.DictReader(open(path)) if 'conv_wino4_kernel' in r['Kernel_Name'] or 'conv_wino4s_kernel' in r['Kernel_Name'] or 'conv_wino4r_kernel' in r['Kernel_Name'] or 'convs_kernel' in r['Kernel_Name'] or 'conv_mfma_kernel' in r['Kernel_Name']
            or 'conv_wino_kernel' in r['Kernel_Name'] or 'conv_wino_res_kernel' in r['Kernel_Name'] or 'conv_wino16_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


def main():
    fpath, wpath, lpath, out = sys.argv[1:5]
    table = json.load(open(lpath))
    layers = table['layers']
    n = table['patches_per_launch']
    f, w = conv_rows(fpath), conv_rows(wpath)
    per = len(layers)
    assert len(f) % per == 0 and len(w) == len(f), (len(f), len(w), per)
    reps = len(f) // per
    res = []
    tot_a = tot_m = 0.0
    for k, L in enumerate(layers):
        fb = sum(float(f[r * per + k]['Counter_Value']) for r in range(reps)) / reps * 1024 * 2     # KiB, x2: gfx950 half-count of wide reads
        wb = sum(float(w[r * per + k]['Counter_Value']) for r in range(reps)) / reps * 1024
        ih, iw, ic = L['in']
        oh, ow, oc = L['out']
        kk = 9 if L['type'] == 'conv3x3' else 4
        frac = L['executed_gflop_per_launch'] / L['algorithmic_gflop_per_launch'] * (4.0 if 'F(4x4)' in L['kernel'] else 2.25 if 'F(2x2)' in L['kernel'] else 1.0)
        alg = 4.0 * n * (ih * iw * ic + oh * ow * oc) * min(frac, 1.0) + 4.0 * kk * ic * oc
        res.append({'layer': L['layer'], 'type': L['type'], 'in': L['in'], 'out': L['out'], 'kernel': L['kernel'],
                    'computed_fraction': round(min(frac, 1.0), 3), 'fetch_GB': round(fb / 1e9, 2), 'write_GB': round(wb / 1e9, 2),
                    'measured_GB': round((fb + wb) / 1e9, 2), 'algorithmic_GB': round(alg / 1e9, 2),
                    'ratio': round((fb + wb) / alg, 2), 'measured_TBs': round((fb + wb) / 1e9 / L['avg_ms'], 2)})
        tot_a += alg; tot_m += fb + wb
    json.dump({'units': 'bytes per launch of %d windows; FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads), WRITE_SIZE KiB x 1024' % n,
               'total_measured_GB': round(tot_m / 1e9, 1), 'total_algorithmic_GB': round(tot_a / 1e9, 1),
               'total_ratio': round(tot_m / tot_a, 2), 'layers': res}, open(out, 'w'), indent=1)
    for r in res:
        print('%-20s %-9s %-16s->%-16s comp %.2f  measured %6.2f GB (%5.2f TB/s)  algorithmic %6.2f GB  x%.2f'
              % (r['layer'], r['type'], r['in'], r['out'], r['computed_fraction'], r['measured_GB'], r['measured_TBs'], r['algorithmic_GB'], r['ratio']))
    print('total measured %.1f GB, algorithmic %.1f GB, x%.2f' % (tot_m / 1e9, tot_a / 1e9, tot_m / tot_a))


if __name__ == '__main__':
    main()
