#!/usr/bin/env python3
"""tests/golden/label_truth_<tag>.npz from the cache of tools/label_truth.py: for the first K full-size images of a seeded
model the HARD pixels - those whose float64 probabilities lie within 2.55e-5 of a point where the quantised argmax changes
(more than the largest error of any float32 evaluator measured: 1.5e-5) - with the float64 label at each of them, plus the
float64 clean-up result in run-length form and n_ec.  On every other pixel all float32 evaluations must agree with each
other; on the hard ones the GPU tests count who is right (tests/test_gpu_configs.py).  Only seeded models ('random',
'smooth'): their weights are regenerated from the seed, nothing else has to travel.

    python tools/make_label_fixture.py random_base64 smooth_base64 [--images 2]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('tags', nargs='+')
    ap.add_argument('--images', type=int, default=2)
    a = ap.parse_args()
    from tools import label_truth
    for tag in a.tags:
        z = np.load(os.path.join(label_truth.CACHE, tag + '.npz'))
        assert str(z['model']) in ('random', 'smooth')
        out = {'model': str(z['model']), 'base': int(z['base']), 'seed0': int(z['seed0']), 'images': a.images,
               'head_gain': label_truth.SMOOTH_HEAD_GAIN if str(z['model']) == 'smooth' else 6.0,
               'margin_unit': 1e-7, 'hard_below': 255, 'shape': np.array(z['raw64'].shape[1:]),
               'oracle32_wrong_px_all_images': (z['raw32'] != z['raw64']).sum(axis=(1, 2)).astype(np.int32),
               'nec64': z['nec64'][:a.images], 'nec32': z['nec32'][:a.images], 'oracle32_max_abs_dp': float(z['p32err'].max())}
        import zlib
        crc, off = [], []
        for i in range(a.images):
            m = z['margin64'][i].ravel()
            idx = np.flatnonzero(m < 255).astype(np.int32)
            out['idx_%d' % i] = idx
            out['truth_%d' % i] = z['raw64'][i].ravel()[idx]
            out['margin_%d' % i] = m[idx]
            # everything OFF the hard pixels in 4 bytes: CRC-32 of the float64 labels with the hard pixels blanked (255).  Every
            # float32 evaluation must reproduce it exactly; `oracle32_off_hard_px` says whether the float32 CPU oracle itself does
            easy = z['raw64'][i].ravel().copy()
            easy[idx] = 255
            crc.append(zlib.crc32(easy.tobytes()) & 0xffffffff)
            o32 = z['raw32'][i].ravel().copy()
            o32[idx] = 255
            off.append(int((o32 != easy).sum()))
        out['crc_easy'] = np.array(crc, np.uint32)
        out['oracle32_off_hard_px'] = np.array(off, np.int32)
        out['oracle32_wrong_on_hard_px'] = np.array([int((z['raw32'][i].ravel()[out['idx_%d' % i]] != out['truth_%d' % i]).sum())
                                                     for i in range(a.images)], np.int32)
        path = os.path.join(ROOT, 'tests', 'golden', 'label_truth_%s.npz' % tag)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), 'bytes;', [len(out['idx_%d' % i]) for i in range(a.images)], 'hard pixels')


if __name__ == '__main__':
    main()
