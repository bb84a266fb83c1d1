#!/usr/bin/env python3
"""Summarise the SQ / GRBM counter passes of tools/profile_round.sh per kernel: matrix-pipe busy fraction, MFMA
instruction counts, effective shader clock.

    python tools/pmc_mfma_summary.py <prof_dir> <out.json>

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over all SIMDs of the chip); SQ_BUSY_CYCLES
/ SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
effective clock of a dispatch is GRBM_GUI_ACTIVE / 8 / its wall time.  v_mfma_f32_32x32x2_f32 occupies a SIMD's matrix
pipe for 64 cycles, so busy cycles = 64 x MFMA instructions when the pipe is never starved.
"""
import collections
import csv
import glob
import json
import sys

N_SIMD = 256 * 4


def load(d):
    files = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)
    if not files:
        return {}
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        k = agg.setdefault(r['Kernel_Name'], collections.defaultdict(float))
        k[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' or 'ns' not in k:
            pass
        key = (r['Dispatch_Id'])
        if key not in k.setdefault('_seen', set()):
            k['_seen'].add(key)
            k['_ns'] = k.get('_ns', 0.0) + float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            k['_launches'] = k.get('_launches', 0) + 1
    return agg


def main():
    d, out = sys.argv[1:3]
    res = {'units': 'see tools/pmc_mfma_summary.py docstring', 'kernels': {}}
    a = load(d + '/pmc_mfma')
    b = load(d + '/pmc_mix')
    tot = collections.defaultdict(float)
    for name, c in a.items():
        if 'conv_' not in name:
            continue
        ns = c['_ns']
        gui = c.get('GRBM_GUI_ACTIVE', 0.0)
        clk = gui / 8.0 / ns if ns else 0.0                        # GHz
        cycles_all_simds = gui / 8.0 * N_SIMD
        e = {'launches': int(c['_launches']), 'total_ms_under_profiler': round(ns / 1e6, 3),
             'effective_clock_GHz': round(clk, 3),
             'SQ_VALU_MFMA_BUSY_CYCLES': c.get('SQ_VALU_MFMA_BUSY_CYCLES'), 'SQ_INSTS_MFMA': c.get('SQ_INSTS_MFMA'),
             'SQ_BUSY_CYCLES': c.get('SQ_BUSY_CYCLES'), 'SQ_WAVE_CYCLES': c.get('SQ_WAVE_CYCLES'), 'GRBM_GUI_ACTIVE': gui,
             'mfma_busy_frac_of_simd_cycles': round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / cycles_all_simds, 4) if cycles_all_simds else None,
             'busy_cycles_per_mfma_inst': round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / c['SQ_INSTS_MFMA'], 2) if c.get('SQ_INSTS_MFMA') else None}
        if name in b:
            for k in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_LDS_BANK_CONFLICT'):
                e[k] = b[name].get(k)
            if b[name].get('SQ_INSTS_VALU') and c.get('SQ_INSTS_MFMA'):
                e['valu_per_mfma_inst'] = round(b[name]['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA'], 2)
        res['kernels'][name] = e
        for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_MFMA', 'GRBM_GUI_ACTIVE', '_ns'):
            tot[k] += c.get(k, 0.0)
        # FLOP per wave-level MFMA instruction: v_mfma_f32_32x32x2_f32 4096, v_mfma_f32_16x16x4_f32 (conv_wino16_kernel) 2048
        tot['_flop'] += c.get('SQ_INSTS_MFMA', 0.0) * (2048.0 if 'conv_wino16_kernel' in name else 4096.0)
    if tot['_ns']:
        cyc = tot['GRBM_GUI_ACTIVE'] / 8.0 * N_SIMD
        res['all_mfma_conv_kernels'] = {'total_ms_under_profiler': round(tot['_ns'] / 1e6, 3),
                                        'effective_clock_GHz': round(tot['GRBM_GUI_ACTIVE'] / 8.0 / tot['_ns'], 3),
                                        'mfma_busy_frac_of_simd_cycles': round(tot['SQ_VALU_MFMA_BUSY_CYCLES'] / cyc, 4),
                                        'executed_tflops_from_SQ_INSTS_MFMA': round(tot['_flop'] / tot['_ns'] / 1e3, 2),
                                        'note': 'executed TFLOP/s = SQ_INSTS_MFMA x FLOP per instruction (v_mfma_f32_32x32x2_f32 4096, conv_wino16_kernel\'s v_mfma_f32_16x16x4_f32 2048) / time; '
                                                'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)'}
    for k in res['kernels'].values():
        pass
    json.dump(res, open(out, 'w'), indent=1, default=lambda o: None)
    print(json.dumps(res.get('all_mfma_conv_kernels')))


if __name__ == '__main__':
    main()
