#!/usr/bin/env python3
"""Randomised whole-pipeline parity: random small U-Nets (depth, width, up-sampling kind, BatchNormalization) x random image
sizes through ecseg_segment_images, checked four ways per case:
  * the cropped plan (default) is bit-identical to itself whatever ran before and however many window lanes it uses, and agrees
    with the uncropped plan to float32 rounding (probabilities within 1e-5; labels differ at most at near-ties of the
    quantised probabilities: its Winograd tiles read zeros outside the receptive field of the pixels the stitch reads);
  * window probabilities vs the CPU oracle within 1e-3 (`predict_on_batch` on the oracle's own tiles);
  * device raw labels differ from the oracle's only at near-ties of the quantised probabilities;
  * clean-up + count are bit-exact functions of the DEVICE raw labels (oracle meta_inference on them);
plus meta_preprocess (uint8 / uint16, gray / RGB) and the overlay row on random FISH images against the oracle.
Runs for --seconds; exit code 1 on any mismatch.

    python tools/fuzz_pipeline.py --seconds 300 [--seed0 0]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=300)
    ap.add_argument('--seed0', type=int, default=0)
    ap.add_argument('--seeds', default=None)
    ap.add_argument('--deep', action='store_true', help='depth up to 4, images up to 1100 x 1500')
    a = ap.parse_args()
    import torch  # noqa: F401
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    from oracle import overlay as o_overlay
    from oracle import pipeline, postproc, preprocess, quant, tiling, unet
    t0 = time.time()
    seed = a.seed0
    todo = [int(x) for x in a.seeds.split(',')] if a.seeds else None
    fails = cases = 0

    def fail(msg):
        nonlocal fails
        fails += 1
        print('FAIL seed %d: %s' % (seed, msg), flush=True)

    while time.time() - t0 < a.seconds:
        if todo is not None:
            if not todo:
                break
            seed = todo.pop(0)
        rng = np.random.default_rng(9 * 10 ** 6 + seed)
        base = int(rng.choice((8, 8, 16, 16, 24, 32)))
        depth = int(rng.choice((3, 4, 4))) if a.deep else int(rng.choice((1, 2, 2, 3)))
        if a.deep:
            base = 8
        up = str(rng.choice(('transpose', 'transpose', 'upsample', 'transpose3', 'transpose4')))
        bn = bool(rng.random() < 0.25)
        cfg = synth.unet_config(base=base, depth=depth, up=up, batchnorm=bn)
        w = synth.unet_weights(cfg, seed=int(rng.integers(0, 1000)))
        H = int(rng.integers(256, 1100 if a.deep else 620)); W = int(rng.integers(256, 1500 if a.deep else 700))
        n = int(rng.integers(1, 3 if a.deep else 4))
        imgs = np.stack([synth.dapi_image(int(rng.integers(0, 10 ** 6)), H, W) for _ in range(n)])
        tag = 'base %d depth %d %s%s, %d x %dx%d' % (base, depth, up, ' +bn' if bn else '', n, H, W)
        try:
            m = MetasegModel(cfg, w, device=0)
            h = m.handle
            # random execution options: none of them may change a single label
            opts = {'images_per_group': int(rng.integers(1, 4)), 'overlap_post': int(rng.integers(0, 2)),
                    'post_graph': int(rng.integers(0, 2)), 'winograd': int(rng.choice((2, 2, 1, 0, 3)))}
            if rng.random() < 0.5:
                opts['post_chunk'] = int(rng.integers(1, 4))
            for kk, vv in opts.items():
                h.set_option(kk, vv)
            tag += ' ' + ','.join('%s=%d' % kv for kv in sorted(opts.items()))
            h.set_option('crop', 1)
            h.set_option('unet_lanes', int(rng.integers(0, 5)))
            raw1, post1, nec1, probs1 = h.segment_images(imgs, want_raw=True, want_probs=True)
            h.set_option('crop', 0)
            raw0, post0, nec0, probs0 = h.segment_images(imgs, want_raw=True, want_probs=True)
            h.set_option('crop', 1)
            h.set_option('unet_lanes', int(rng.integers(0, 5)))
            raw2, post2, nec2, probs2 = h.segment_images(imgs, want_raw=True, want_probs=True)
            if not (np.array_equal(raw1, raw2) and np.array_equal(post1, post2) and np.array_equal(nec1, nec2) and np.array_equal(probs1, probs2)):
                fail('cropped plan differs from itself after another run / with other lanes (%d raw px) - %s' % (int((raw1 != raw2).sum()), tag))
            if float(np.abs(probs1 - probs0).max()) > 1e-5:
                fail('crop on/off probabilities differ by %.3e - %s' % (float(np.abs(probs1 - probs0).max()), tag))
            dc = raw1 != raw0
            if dc.any():
                top = np.sort(quant.quantise_u8(probs0.astype(np.float64)).astype(np.int32), axis=-1)
                if ((top[..., 3] - top[..., 2])[dc] > 1).any():
                    fail('crop on/off labels differ away from ties (%d raw px) - %s' % (int(dc.sum()), tag))
            pos = tiling.patch_positions(H, W)
            k = int(rng.integers(0, n))
            patches = tiling.extract_patches(imgs[k][..., None], pos)
            want_p = unet.forward(cfg, w, patches)
            got_p = m.predict_on_batch(patches)
            err = float(np.abs(got_p - want_p).max())
            if not np.isfinite(got_p).all() or err > 1e-3:
                fail('probabilities differ by %.3e - %s' % (err, tag))
            want_raw = pipeline.raw_labels_from_probs(want_p, pos)
            d = raw1[k] != want_raw
            if d.any():
                q = quant.quantise_u8(tiling.stitch(want_p, pos)).astype(np.int32)
                top = np.sort(q, axis=2)
                if ((top[..., 3] - top[..., 2])[d] > 1).any():                 # not a near-tie of the quantised probabilities
                    fail('raw labels differ away from ties (%d px) - %s' % (int(d.sum()), tag))
            for j in range(n):
                want_post = postproc.meta_inference(raw1[j])
                if not np.array_equal(post1[j], want_post) or int(nec1[j]) != postproc.count_cc(want_post == 3)[0]:
                    fail('clean-up / count differ on the device raw labels - %s' % tag)
            # meta_preprocess on a random container of the same pixels
            kind = int(rng.integers(0, 4))
            src = imgs if kind < 2 else (imgs.astype(np.uint16) * int(rng.choice((1, 16, 257))))
            if kind % 2:
                src = np.stack([src // 2, src // 3, src], axis=-1).astype(src.dtype)
            if rng.random() < 0.5:
                src = src.max() - src                                             # bright background: the inversion branch
            g, inv = h.preprocess(src)
            for j in range(n):
                if not np.array_equal(g[j], preprocess.meta_preprocess(src[j])):
                    fail('meta_preprocess differs (%s %s) - %s' % (src.dtype, src.shape, tag))
            # overlay row on random FISH channels over the cleaned labels
            rgb = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
            rgb[rng.random((n, H, W)) < 0.6] = 0
            sens = int(rng.integers(0, 255))
            rows = h.overlay(post1, rgb, sens)
            for j in range(n):
                want = o_overlay.overlay_row(post1[j].astype(np.int64), rgb[j], sens)
                flat = []
                for v in want:
                    flat += [int(v[0]), -1 if isinstance(v[1], float) else int(v[1])] if isinstance(v, tuple) else [int(v)]
                if [int(x) for x in np.asarray(rows[j]).ravel()] != flat:
                    fail('overlay row differs - %s' % tag)
            del m
        except Exception as e:
            fail('%s: %s - %s' % (type(e).__name__, e, tag))
        cases += 1
        seed += 1
    print('pipeline fuzz: %d cases, %d failure(s), %.0f s' % (cases, fails, time.time() - t0), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
