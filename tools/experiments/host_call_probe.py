#!/usr/bin/env python3
"""Where does a `make metaseg` device call spend its time?  One batch of raw RGB images through
  A  ecseg_preprocess + ecseg_segment_images_ex, pageable host arrays (rounds 1-4)
  B  ecseg_meta_segment, pageable host arrays
  C  ecseg_meta_segment, page-locked input and outputs (ecseg_host_alloc)
  D  C from two threads with a handle each (config key device_workers = 2)
  E  C on ONE handle with the next batch's images sent ahead (ecseg_prefetch_input)
against the device-only time of the same batch (stage timers).  Prints one JSON line.

    python tools/experiments/host_call_probe.py [--base 16] [--batch 32] [--reps 6]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--base', type=int, default=16)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=6)
    a = ap.parse_args()
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    cfg = synth.unet_config(base=a.base)
    models = [MetasegModel(cfg, synth.unet_weights(cfg, seed=0), device=0) for _ in range(2)]
    if a.base >= 64:
        for m in models:
            m.handle.set_images_per_group(16)
    base = [synth.dapi_image(600 + i, rgb=True) for i in range(8)]
    imgs = np.stack([np.roll(base[i % 8], (31 * (i // 8), 17 * (i // 8)), axis=(0, 1)) for i in range(a.batch)])
    n, H, W, _ = imgs.shape
    out = {'unet_base': a.base, 'batch_images': a.batch, 'reps': a.reps, 'image_shape': list(imgs.shape[1:]), 'dtype': str(imgs.dtype)}

    def timed(fn, reps=a.reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return round(1e3 * float(np.median(ts)) / n, 4)

    h = models[0].handle
    ref = {}

    def two_calls():
        gray, _ = h.preprocess(imgs)
        r = models[0].segment_ex(gray)
        ref['gray'], ref['post'], ref['nec'] = gray, r[0], r[1]

    out['A_two_calls_pageable_ms_per_image'] = timed(two_calls)
    out['device_only_ms_per_image'] = round(sum(h.timings().values()) / n, 4)

    got = {}

    def fused():
        got['r'] = h.meta_segment(imgs)

    out['B_fused_pageable_ms_per_image'] = timed(fused)
    out['B_identical_to_A'] = bool(np.array_equal(got['r'][0], ref['gray']) and np.array_equal(got['r'][1], ref['post'])
                                   and np.array_equal(got['r'][2], ref['nec']))

    pin = []
    for m in models:
        hh = m.handle
        t0 = time.perf_counter()
        p_in = hh.host_empty(imgs.shape, imgs.dtype)
        p_g, p_p = hh.host_empty((n, H, W)), hh.host_empty((n, H, W))
        out.setdefault('pinned_alloc_ms', []).append(round(1e3 * (time.perf_counter() - t0), 2))
        p_in[...] = imgs
        pin.append((hh, p_in, p_g, p_p))

    def fused_pinned(k=0):
        hh, p_in, p_g, p_p = pin[k]
        got['p%d' % k] = hh.meta_segment(p_in, gray_out=p_g, post_out=p_p)

    out['C_fused_pinned_ms_per_image'] = timed(fused_pinned)
    out['C_identical_to_A'] = bool(np.array_equal(got['p0'][0], ref['gray']) and np.array_equal(got['p0'][1], ref['post'])
                                   and np.array_equal(got['p0'][2], ref['nec']))
    fused_pinned(1)

    def two_workers():
        th = [threading.Thread(target=fused_pinned, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    out['D_two_handles_pinned_ms_per_image'] = round(timed(two_workers) / 2, 4)        # (2 n images per repetition)

    # E: ONE handle, the next batch's images sent ahead under this batch's kernels (ecseg_prefetch_input)
    hh, p_in, p_g, p_p = pin[0]
    p_in_b = hh.host_empty(imgs.shape, imgs.dtype)
    p_in_b[...] = imgs
    bufs = [p_in, p_in_b]

    def ahead():
        for k in range(4):
            hh.prefetch_input(bufs[(k + 1) & 1])
            got['e'] = hh.meta_segment(bufs[k & 1], gray_out=p_g, post_out=p_p)

    out['E_one_handle_inputs_sent_ahead_ms_per_image'] = round(timed(ahead) / 4, 4)
    out['E_identical_to_A'] = bool(np.array_equal(got['e'][0], ref['gray']) and np.array_equal(got['e'][1], ref['post'])
                                   and np.array_equal(got['e'][2], ref['nec']))
    t0 = time.perf_counter()
    c = np.empty_like(pin[0][1])
    np.copyto(c, imgs)
    out['host_copy_of_the_batch_ms_per_image'] = round(1e3 * (time.perf_counter() - t0) / n, 4)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
