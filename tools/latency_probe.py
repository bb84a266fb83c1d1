#!/usr/bin/env python3
"""Single-image latency (BASELINE configs[1] read literally) with its stage breakdown, for a few option sets, and the same
for small batches (2, 3 images) - the window lanes of run_plan (api.hip) are what these measure.
    python tools/latency_probe.py [--base 64] [--lanes 1,2,3,4] [--images 1,2,3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--base', type=int, default=64)
    ap.add_argument('--lanes', default='1,2,3,4')
    ap.add_argument('--images', default='1,2,3')
    ap.add_argument('--weights', default='random', choices=('random', 'smooth'),
                    help='smooth: the fitted base-16 stand-in for a trained model (tests/golden/smooth_b16_f16.npz; implies --base 16)')
    a = ap.parse_args()
    import torch
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    if a.weights == 'smooth':
        from tools import make_smooth_fixture
        cfg, weights = make_smooth_fixture.load(os.path.join(ROOT, 'tests', 'golden', 'smooth_b16_f16.npz'))
    else:
        cfg = synth.unet_config(base=a.base)
        weights = synth.unet_weights(cfg, seed=0)
    m = MetasegModel(cfg, weights, device=0)
    h = m.handle
    out = {}
    for n_img in [int(x) for x in a.images.split(',')]:
        img = torch.from_numpy(np.stack([synth.dapi_image(3 + i) for i in range(n_img)])).cuda()
        raw, post = torch.empty_like(img), torch.empty_like(img)
        nec = torch.zeros(n_img, dtype=torch.int32, device='cuda')
        call = lambda: h.segment_images_dev(img.data_ptr(), n_img, 1040, 1392, raw.data_ptr(), post.data_ptr(), nec.data_ptr())
        ref = None
        variants = [('lanes%d' % int(x), {'unet_lanes': int(x)}) for x in a.lanes.split(',')]
        if n_img == 1:
            variants += [('auto', {}), ('auto+post_graph', {'post_graph': 1}), ('auto+overlap_post', {'overlap_post': 1})]
        for name, opts in variants:
            for k in ('post_graph', 'overlap_post', 'unet_lanes'):
                h.set_option(k, opts.get(k, 0))
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            got = (raw.cpu().numpy().copy(), post.cpu().numpy().copy(), nec.cpu().numpy().copy())
            if ref is None:
                ref = got
            same = all(np.array_equal(x, y) for x, y in zip(ref, got))
            ts, st = [], []
            for _ in range(10):
                t0 = time.perf_counter()
                call()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                st.append(h.timings())
            out['%dimg_%s' % (n_img, name)] = {'median_ms': round(float(np.median(ts)), 3), 'min_ms': round(min(ts), 3),
                                               'identical_to_first_variant': bool(same),
                                               'stages_ms': {k: round(float(np.median([s[k] for s in st])), 3) for k in st[0]}}
            print(name, n_img, out['%dimg_%s' % (n_img, name)], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
