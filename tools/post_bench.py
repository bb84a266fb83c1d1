#!/usr/bin/env python3
"""meta_inference + count ("CCL ms/image") on the device, by input kind: synthetic label maps of realistic scenes
(synth.label_map: blobs + 0.2 % salt noise), the raw argmax output of a fitted smooth-output U-Net, and the speckled raw
output of the random-weight bench model (worst case for union-find: ~17 k components per image).  One JSON line.

    python tools/post_bench.py [--images 64] [--reps 5]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W = 1040, 1392
ALG_BYTES_PER_IMAGE = 7 * H * W * 9 + 7 * H * W * 2          # 7 labellings x 13.0 MB + 7 stencils x 2.9 MB


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=64)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--base', type=int, default=16)
    ap.add_argument('--opt', action='append', default=[], help='library option key=value')
    ap.add_argument('--only', default=None, help='time only the input kinds whose name contains this text (per-kind rocprofv3 runs)')
    a = ap.parse_args()
    import torch
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    from tools import fit_smooth_model
    n = a.images
    cfg_f, w_f = fit_smooth_model.fit(base=a.base, steps=150, threads=8)          # CPU, before the GPU is touched
    kinds = {}
    base_maps = [synth.label_map(i) for i in range(16)]
    kinds['synth.label_map (blobs + 0.2 % salt noise)'] = np.stack([np.roll(base_maps[i % 16], (13 * (i // 16), 29 * (i // 16)), axis=(0, 1)) for i in range(n)])
    imgs = np.stack([synth.dapi_image(i) for i in range(16)])
    for tag, cfg, w in (('raw argmax of a fitted (smooth-output) base-%d U-Net' % a.base, cfg_f, w_f),
                        ('raw argmax of the random-weight base-64 bench model (speckle)', synth.unet_config(base=64), None)):
        if w is None:
            w = synth.unet_weights(cfg, seed=0)
        m = MetasegModel(cfg, w, device=0)
        raw, _, _ = m.handle.segment_images(imgs, want_raw=True)
        kinds[tag] = np.stack([np.roll(raw[i % 16], (13 * (i // 16), 29 * (i // 16)), axis=(0, 1)) for i in range(n)])
        hnd = m.handle
    out = {'images_per_call': n, 'image_size': [H, W], 'algorithmic_bytes_per_image': ALG_BYTES_PER_IMAGE, 'hbm_peak_GBs': 8000.0, 'inputs': {}}
    for tag, lab in kinds.items():
        if a.only and a.only not in tag:
            continue
        d_in = torch.from_numpy(np.ascontiguousarray(lab)).cuda()
        d_out = torch.empty_like(d_in)
        nec = torch.zeros(n, dtype=torch.int32, device='cuda')
        hnd.set_option('post_chunk', n)
        for kv in a.opt:
            hnd.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        hnd.meta_inference_dev(d_in.data_ptr(), n, H, W, d_out.data_ptr(), nec.data_ptr())
        ms = []
        for _ in range(a.reps):
            hnd.meta_inference_dev(d_in.data_ptr(), n, H, W, d_out.data_ptr(), nec.data_ptr())
            ms.append(hnd.timings()['post'])
        t = float(np.median(ms)) / n
        out['inputs'][tag] = {'ms_per_image': round(t, 4), 'achieved_GBs': round(ALG_BYTES_PER_IMAGE / (t * 1e-3) / 1e9, 1),
                              'frac_of_hbm_peak': round(ALG_BYTES_PER_IMAGE / (t * 1e-3) / 8e12, 4),
                              'mean_n_ec': float(nec.float().mean().item()),
                              'class_fractions': [round(float(v), 4) for v in np.bincount(lab.ravel(), minlength=4) / lab.size]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
