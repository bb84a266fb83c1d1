#!/usr/bin/env python3
"""Real-weights validation kit.  Neither TensorFlow nor models/metaseg.h5 exists in the build
container, so the Keras forward pass is "parity unpinned" there; this script closes that gap wherever both exist.

Step 1 (reference environment: TF 2.8 + the ecSeg conda env, CPU is fine):
    python tools/validate_real_weights.py dump models/metaseg.h5 example_ecSeg/input.tif keras_ref.npz
        -> tiles the image exactly like src/utils.py:113, runs model.predict_on_batch, stores patches + probabilities.

Step 2 (MI355X box with this repository built):
    python tools/validate_real_weights.py check models/metaseg.h5 keras_ref.npz
        -> reports max |p_hip - p_keras| (bar: <= 1e-3), the number of pixels whose quantised argmax differs, and the
           layer types found in the file's model_config.
"""
import json
import sys

import numpy as np


def dump(h5, image, out):
    import tensorflow as tf
    from skimage.io import imread
    sys.path.insert(0, 'src')
    from image_tools import im2patches_overlap, meta_preprocess      # the reference's own functions
    model = tf.keras.models.load_model(h5)
    img = meta_preprocess(imread(image))
    _, patches, pos = im2patches_overlap(np.expand_dims(img, -1))
    x = np.array(patches)
    np.savez_compressed(out, patches=x, pos=np.array(pos), probs=model.predict_on_batch(x), gray=img)
    print('wrote', out, x.shape)


def check(h5, ref):
    from collections import Counter
    from ecseg_amd import hdf5_min
    from ecseg_amd.model import MetasegModel
    cfg, _ = hdf5_min.load_keras_h5(h5)
    kinds = Counter(L['class_name'] for L in json.loads(cfg)['config']['layers'])
    print('layers:', dict(kinds))
    r = np.load(ref)
    model = MetasegModel.from_h5(h5)
    got = model.predict_on_batch(r['patches'])
    err = float(np.abs(got - r['probs']).max())
    q = lambda p: np.argmax(np.clip(np.rint(p.astype(np.float64) * 255), 0, 255), -1)
    diff = int((q(got) != q(r['probs'])).sum())
    print('max |p_hip - p_keras| = %.3e (bar 1e-3); quantised-argmax mismatches on patch pixels: %d of %d'
          % (err, diff, got[..., 0].size))
    post, nec = model.segment(r['gray'])
    print('n_ec on the stitched image:', nec)
    sys.exit(0 if err <= 1e-3 else 1)


if __name__ == '__main__':
    if len(sys.argv) >= 5 and sys.argv[1] == 'dump':
        dump(*sys.argv[2:5])
    elif len(sys.argv) >= 4 and sys.argv[1] == 'check':
        check(*sys.argv[2:4])
    else:
        print(__doc__)
        sys.exit(2)
