#!/usr/bin/env python3
"""Cache the smooth-output base-16 U-Net that tests/test_gpu_configs.py::test_smooth_output_model_labels_vs_cpu_oracle uses
(VERDICT r03 item 8): the weights of tools/fit_smooth_model.fit(base=16, steps=120) rounded to float16 ->
tests/golden/smooth_b16_f16.npz (3.7 MB).  The test casts them back to float32 - device and oracle see the same float32
numbers; nothing depends on the torch build that made them, and no fit runs inside the GPU test session any more.

    python tools/make_smooth_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tools import fit_smooth_model
    cfg, w = fit_smooth_model.fit(base=16, steps=120, threads=8)
    arrays = {}
    for name, arrs in w.items():
        for i, a in enumerate(arrs):
            arrays['%s/%d' % (name, i)] = np.asarray(a, np.float16)
    out = os.path.join(ROOT, 'tests', 'golden', 'smooth_b16_f16.npz')
    np.savez_compressed(out, **arrays)
    print(out, os.path.getsize(out))


def load(path):
    """-> (model_config, weights) with float32 arrays."""
    from ecseg_amd import synth
    cfg = synth.unet_config(base=16)
    z = np.load(path)
    w = {}
    for key in z.files:
        name, i = key.rsplit('/', 1)
        w.setdefault(name, {})[int(i)] = z[key].astype(np.float32)
    return cfg, {name: [d[i] for i in sorted(d)] for name, d in w.items()}


if __name__ == '__main__':
    main()
