#!/usr/bin/env python3
"""Which kernel family each convolution of the canonical U-Net takes, per base width, with per-layer time (bench shapes:
16 images x 35 windows).  python tools/kernel_map.py [bases...] -> JSON on stdout."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import bench
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    bases = [int(b) for b in sys.argv[1:]] or [64, 32, 16]
    out = {}
    imgs = np.stack([synth.dapi_image(i) for i in range(16)])
    for base in bases:
        cfg = synth.unet_config(base=base)
        m = MetasegModel(cfg, synth.unet_weights(cfg, seed=0), device=0)
        res = {}
        for mode in (2, 1):
            m.handle.set_option('winograd', mode)
            m.handle.segment_images(imgs, want_raw=False)
            m.handle.set_kernel_profiling(True)
            recs = []
            for _ in range(3):
                m.handle.segment_images(imgs, want_raw=False)
                recs += m.handle.conv_launch_profile()
            m.handle.set_kernel_profiling(False)
            rows = bench.layer_table(m, recs, 3)
            t = m.handle.timings()
            res['winograd=%d' % mode] = {'unet_ms_per_image': round(t['unet'] / 16, 3), 'mfma_conv_ms_per_step': round(sum(r['avg_ms'] for r in rows), 3),
                                         'layers': [{k: r[k] for k in ('layer', 'type', 'in', 'out', 'kernel', 'avg_ms', 'executed_frac_of_peak')} for r in rows]}
        out['base_%d' % base] = res
        m.handle.close()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
