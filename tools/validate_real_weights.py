#!/usr/bin/env python3
"""Real-weights validation kit.  Neither TensorFlow nor models/metaseg.h5 exists in the build
container, so the Keras forward pass is "parity unpinned" there; this script closes that gap wherever both exist, all the
way to `north_star`'s bar: labels and ec_quantification.csv of the reference run.

Step 1 (reference environment: TF 2.8 + the ecSeg conda env, CPU is fine; never on the GPU box):
    python tools/validate_real_weights.py dump models/metaseg.h5 example_ecSeg/input.tif keras_ref.npz
        -> runs the reference's own meta_segment steps (src/utils.py:109-120) with its own functions and stores every stage:
           pre-processed image, patches + positions, Keras probabilities, the stitched canvas, the uint8-quantised argmax
           labels, the labels after meta_inference, count_cc(I == 3)[0] and the CSV text the reference writes
           (src/metaseg.py:44-57).
    (`dump --oracle ...` lets this repository's CPU oracle stand in for TensorFlow: it exercises the kit itself - used by
    tests/test_gpu_more.py on the synthetic .h5 fixture - and says nothing about TensorFlow.)

Step 2 (MI355X box with this repository built):
    python tools/validate_real_weights.py check models/metaseg.h5 keras_ref.npz
        -> per stage: max |p_hip - p_ref| on the patches (bar 1e-3), raw-label pixels that differ and how many of those are NOT
           tie-risk pixels (two largest quantised probabilities more than 1 apart: must be 0 - anything else is a real
           difference, not float32 rounding), final-label pixels that differ, n_ec, and the CSV text diff.  Exit code 0 only
           when probabilities meet the bar, no mismatch lies outside the tie-risk set, and labels + CSV are identical; 3 when
           only tie-risk pixels differ (float32 summation order), 1 otherwise.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _csv_text(name, n_ec):
    # pandas DataFrame(columns=['image_name', '# of ec']).to_csv(index=False) of one row (src/metaseg.py:44-57)
    return 'image_name,# of ec\n%s,%d\n' % (name, n_ec)


def dump(h5, image, out, use_oracle=False):
    name = os.path.split(image)[1]
    if use_oracle:
        sys.path.insert(0, ROOT)
        from ecseg_amd import hdf5_min, image_io
        from oracle import pipeline, postproc, preprocess, quant, tiling, unet
        cfg, weights = hdf5_min.load_keras_h5(h5)
        img = preprocess.meta_preprocess(image_io.imread(image))
        pos = tiling.patch_positions(*img.shape)
        x = tiling.extract_patches(img[..., None], pos)
        probs = unet.forward(cfg, weights, x)
        canvas = tiling.stitch(probs, pos)
        raw = quant.quantised_argmax(canvas)
        final = postproc.meta_inference(raw)
        n_ec = int(postproc.count_cc(final == 3)[0])
        del pipeline
    else:
        import tensorflow as tf
        from skimage import img_as_ubyte
        from skimage.io import imread
        sys.path.insert(0, 'src')
        from image_tools import count_cc, im2patches_overlap, meta_inference, meta_preprocess, patches2im_overlap   # the reference's own functions
        model = tf.keras.models.load_model(h5)
        img = meta_preprocess(imread(image))
        dim, patches, pos = im2patches_overlap(np.expand_dims(img, -1))
        x = np.array(patches)
        probs = model.predict_on_batch(x)
        canvas = patches2im_overlap(probs, pos, dim)                            # src/utils.py:116
        raw = np.argmax(img_as_ubyte(canvas), axis=-1)                          # src/utils.py:117-118
        final = meta_inference(raw.copy())                                      # src/utils.py:119
        n_ec = int(count_cc(final == 3)[0])                                     # src/metaseg.py:46
    np.savez_compressed(out, patches=x, pos=np.array(pos), probs=np.asarray(probs, np.float32), gray=img,
                        canvas=np.asarray(canvas, np.float32), raw=np.asarray(raw, np.uint8), final=np.asarray(final, np.uint8),
                        n_ec=np.int64(n_ec), csv=np.array(_csv_text(name, n_ec)), image_name=np.array(name),
                        source=np.array('oracle' if use_oracle else 'tensorflow'))
    print('wrote', out, x.shape, 'n_ec', n_ec)


def check(h5, ref, precision='fast'):
    from collections import Counter
    sys.path.insert(0, ROOT)
    from ecseg_amd import hdf5_min
    from ecseg_amd.model import MetasegModel
    cfg, _ = hdf5_min.load_keras_h5(h5)
    kinds = Counter(L['class_name'] for L in (json.loads(cfg) if isinstance(cfg, (str, bytes)) else cfg)['config']['layers'])
    print('layers:', dict(kinds))
    r = np.load(ref)
    print('reference side:', str(r['source']) if 'source' in r else 'tensorflow (old dump)')
    model = MetasegModel.from_h5(h5)
    model.handle.set_option('winograd', 1 if precision == 'exact' else 2)
    got = model.predict_on_batch(r['patches'])
    err = float(np.abs(got - r['probs']).max())
    print('max |p_hip - p_ref| = %.3e (bar 1e-3)' % err)
    ok_p = err <= 1e-3
    if 'raw' not in r:                                                          # a dump of the round-3 kit: probabilities only
        post, nec = model.segment(r['gray'])
        print('n_ec on the stitched image:', nec, '(the dump holds no labels: re-run `dump` with this version)')
        sys.exit(0 if ok_p else 1)
    raw, post, nec, tie, probs = model.handle.segment_images(r['gray'], want_raw=True, want_tie_risk=True, want_probs=True)
    raw, post, nec, probs = raw[0], post[0], int(nec[0]), probs[0]
    q = np.sort(np.clip(np.rint(np.asarray(r['canvas'], np.float64) * 255.0), 0, 255), axis=-1)
    risky = (q[..., 3] - q[..., 2] <= 1)                                        # on the REFERENCE's own quantised values
    d_raw = raw != r['raw']
    outside = int((d_raw & ~risky).sum())
    d_final = int((post != r['final']).sum())
    text = _csv_text(str(r['image_name']), nec)
    print('stitched probabilities: max |dp| = %.3e' % float(np.abs(probs - r['canvas']).max()))
    print('raw argmax labels: %d of %d pixels differ; %d of them outside the reference\'s tie-risk set (%d pixels; device count %d)'
          % (int(d_raw.sum()), raw.size, outside, int(risky.sum()), int(tie[0])))
    print('labels after meta_inference: %d pixels differ' % d_final)
    print('n_ec: device %d, reference %d' % (nec, int(r['n_ec'])))
    same_csv = text == str(r['csv'])
    print('ec_quantification.csv: %s' % ('identical' if same_csv else 'DIFFERS\n--- reference\n%s--- device\n%s' % (str(r['csv']), text)))
    if ok_p and not d_raw.any() and d_final == 0 and same_csv:
        sys.exit(0)
    sys.exit(3 if ok_p and outside == 0 else 1)


if __name__ == '__main__':
    a = [x for x in sys.argv[1:] if not x.startswith('--')]
    flags = [x for x in sys.argv[1:] if x.startswith('--')]
    if len(a) >= 4 and a[0] == 'dump':
        dump(a[1], a[2], a[3], use_oracle='--oracle' in flags)
    elif len(a) >= 3 and a[0] == 'check':
        check(a[1], a[2], precision='exact' if '--exact' in flags else 'fast')
    else:
        print(__doc__)
        sys.exit(2)
