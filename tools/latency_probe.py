#!/usr/bin/env python3
"""Single-image latency (BASELINE configs[1] read literally) with its stage breakdown, for a few option sets.
    python tools/latency_probe.py [--base 64]"""
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
    a = ap.parse_args()
    import torch
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    cfg = synth.unet_config(base=a.base)
    m = MetasegModel(cfg, synth.unet_weights(cfg, seed=0), device=0)
    h = m.handle
    img = torch.from_numpy(synth.dapi_image(3)[None]).cuda()
    raw, post = torch.empty_like(img), torch.empty_like(img)
    nec = torch.zeros(1, dtype=torch.int32, device='cuda')
    out = {}
    for name, opts in (('default', {}), ('post_graph', {'post_graph': 1}), ('overlap_post', {'overlap_post': 1})):
        for k in ('post_graph', 'overlap_post'):
            h.set_option(k, opts.get(k, 0))
        call = lambda: h.segment_images_dev(img.data_ptr(), 1, 1040, 1392, raw.data_ptr(), post.data_ptr(), nec.data_ptr())
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts, st = [], []
        for _ in range(10):
            t0 = time.perf_counter()
            call()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            st.append(h.timings())
        out[name] = {'median_ms': round(float(np.median(ts)), 3),
                     'stages_ms': {k: round(float(np.median([s[k] for s in st])), 3) for k in st[0]}}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
