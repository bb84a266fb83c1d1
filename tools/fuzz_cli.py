#!/usr/bin/env python3
"""Randomised file-level parity of the `make metaseg` / `make meta_overlay` loops: folders of images of MIXED sizes, sample
types and TIFF flavours (gray / RGB, uint8 / uint16, none / LZW / deflate / PackBits, written by libtiff through PIL) and
`.npy` inputs, random batch sizes and I/O thread counts.  Checked per image: `labels/<stem>.npy` (int64) equals a direct
single-image device call on the pixels PIL decodes; `dapi/<name>` read back by PIL is the inverted pre-processed image;
`labels/<stem>.png` has the reference's four colours in the places of the labels; records come back in path order with the
count of that image; overlay rows equal the oracle's on the stored labels.  Runs for --seconds; exit code 1 on any mismatch.

    python tools/fuzz_cli.py --seconds 200 [--seed0 0]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COLORS = np.array([[0x38, 0x6c, 0xb0], [0xff, 0xff, 0x99], [0x7f, 0xc9, 0x7f], [0xf0, 0x02, 0x7f]], np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=200)
    ap.add_argument('--seed0', type=int, default=0)
    ap.add_argument('--seeds', default=None)
    a = ap.parse_args()
    from PIL import Image
    import torch  # noqa: F401
    from ecseg_amd import dist, meta_overlay, metaseg, synth
    from ecseg_amd.model import MetasegModel
    from oracle import overlay as o_overlay
    cfg = synth.unet_config(base=8, depth=2)
    model = MetasegModel(cfg, synth.unet_weights(cfg, seed=5), device=0)
    h = model.handle
    t0 = time.time()
    seed = a.seed0
    todo = [int(x) for x in a.seeds.split(',')] if a.seeds else None
    fails = cases = n_files = 0

    def fail(msg):
        nonlocal fails
        fails += 1
        print('FAIL seed %d: %s' % (seed, msg), flush=True)

    while time.time() - t0 < a.seconds:
        if todo is not None:
            if not todo:
                break
            seed = todo.pop(0)
        rng = np.random.default_rng(7 * 10 ** 6 + seed)
        d = tempfile.mkdtemp(prefix='ecseg_fuzz_cli_')
        try:
            shapes = [(int(rng.integers(256, 400)), int(rng.integers(256, 420))) for _ in range(int(rng.integers(1, 4)))]
            paths, pixels = [], {}
            for k in range(int(rng.integers(3, 12))):
                H, W = shapes[int(rng.integers(0, len(shapes)))]
                g = synth.dapi_image(int(rng.integers(0, 10 ** 6)), H, W)
                kind = int(rng.integers(0, 6))
                name = '%s_%02d' % (''.join(rng.choice(list('abcXYZ019_-'), size=int(rng.integers(1, 8)))), k)
                if kind == 5:                                              # .npy input (src/utils.py:105-107 globs them too)
                    p = os.path.join(d, name + '.npy')
                    np.save(p, g)
                    arr = g
                else:
                    p = os.path.join(d, name + '.tif')
                    comp = [None, 'tiff_lzw', 'tiff_adobe_deflate', 'packbits'][int(rng.integers(0, 4))]
                    if kind in (0, 1):
                        arr = g
                    elif kind == 2:
                        arr = np.stack([g // 3, g // 2, g], axis=-1)       # blue = DAPI
                    else:
                        arr = (g.astype(np.uint16) * int(rng.choice((1, 16, 257))))
                    if rng.random() < 0.3:
                        arr = arr.max() - arr                              # bright background
                    Image.fromarray(arr).save(p, compression=comp)
                paths.append(p)
                pixels[p] = arr
            paths = sorted(p for p in paths if p.endswith('.tif')) + sorted(p for p in paths if p.endswith('.npy'))
            for sub in ('dapi', 'labels', 'red', 'green'):
                os.makedirs(os.path.join(d, sub), exist_ok=True)
            bi, it = int(rng.integers(1, 6)), int(rng.integers(1, 7))
            # (page-locked batch buffers + inputs sent ahead, forced on for half of the folders: small folders would skip them)
            rec = metaseg.run(d, model, paths, batch_images=bi, io_threads=it, log=lambda *x: None,
                              pinned_min_images=0 if rng.random() < 0.5 else None)
            tag = '%d files, %d shapes, batch %d, threads %d' % (len(paths), len(shapes), bi, it)
            if len(rec) != len(paths) or [int(r[dist.F_INDEX]) for r in rec] != list(range(len(paths))):
                fail('records out of order / missing - ' + tag)
            for k, p in enumerate(paths):
                gray, inv = h.preprocess(pixels[p][None])
                raw, post, nec = h.segment_images(gray, want_raw=False)
                lab = np.load(os.path.join(d, 'labels', os.path.basename(p)[:-4] + '.npy'))
                if lab.dtype != np.int64 or not np.array_equal(lab, post[0]):
                    fail('labels of %s differ from the direct device call - %s' % (os.path.basename(p), tag))
                if int(rec[k][dist.F_STATUS]) != 0 or int(rec[k][dist.F_NEC]) != int(nec[0]):
                    fail('record of %s: status %d n_ec %d, direct call %d - %s' % (os.path.basename(p), rec[k][dist.F_STATUS], rec[k][dist.F_NEC], nec[0], tag))
                dp = os.path.join(d, 'dapi', os.path.basename(p))
                back = np.load(dp) if dp.endswith('.npy') else np.array(Image.open(dp))
                if not np.array_equal(back, ~gray[0]):
                    fail('dapi/%s is not the inverted pre-processed image - %s' % (os.path.basename(p), tag))
                png = np.array(Image.open(os.path.join(d, 'labels', os.path.basename(p)[:-4] + '.png')).convert('RGB'))
                if not np.array_equal(png, COLORS[post[0]]):
                    fail('labels PNG colours of %s - %s' % (os.path.basename(p), tag))
                n_files += 1
            rgbs = [p for p in paths if pixels[p].ndim == 3]
            if rgbs:
                sens = int(rng.integers(0, 200))
                rows = meta_overlay.run(d, h, rgbs, sens, batch_images=bi, io_threads=it, log=lambda *x: None)
                if [r[0] for r in rows] != [os.path.basename(p) for p in rgbs]:
                    fail('overlay rows out of order - ' + tag)
                for r, p in zip(rows, rgbs):
                    lab = np.load(os.path.join(d, 'labels', os.path.basename(p)[:-4] + '.npy'))
                    want = o_overlay.overlay_row(lab, pixels[p], sens)
                    if o_overlay.csv_text(['x'] * 10, [r]) != o_overlay.csv_text(['x'] * 10, [[r[0]] + want]):
                        fail('overlay row of %s - %s' % (os.path.basename(p), tag))
        except Exception as e:
            import traceback
            traceback.print_exc()
            fail('%s: %s' % (type(e).__name__, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        cases += 1
        seed += 1
    print('cli fuzz: %d folders, %d files, %d failure(s), %.0f s' % (cases, n_files, fails, time.time() - t0), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
