"""Per-kernel totals of the float64 adjudicator over the 32 fixture images (tests/golden/label_truth_*.npz): what the in-suite bound of
tests/test_gpu_configs.py::test_labels_vs_float64_adjudicator is derived from (measured totals + 25 %)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
from ecseg_amd import keras_plan, synth  # noqa: E402
from ecseg_amd._lib import Handle  # noqa: E402

out = {}
for tag in ('random_base64', 'smooth_base64'):
    z = np.load(os.path.join('tests', 'golden', 'label_truth_%s.npz' % tag))
    cfg = synth.unet_config(base=int(z['base']))
    w = synth.unet_weights(cfg, seed=0, smooth=str(z['model']) == 'smooth', head_gain=float(z['head_gain']))
    n = int(z['images'])
    imgs = np.stack([synth.dapi_image(int(z['seed0']) + i) for i in range(n)])
    h = Handle(0)
    h.load_plan(keras_plan.build_plan(cfg, w))
    res = {'oracle32_wrong': int(z['oracle32_wrong_on_hard_px'].sum())}
    for mode in (3, 2, 1, 0):
        h.set_option('winograd', mode)
        raw, post, nec, tie = h.segment_images(imgs, want_raw=True, want_tie_risk=True)
        per = [int((raw[i].ravel()[z['idx_%d' % i].astype(np.int64)] != z['truth_%d' % i]).sum()) for i in range(n)]
        res['mode%d' % mode] = {'total': sum(per), 'worst_image': max(per), 'min_tie_risk': int(tie.min()), 'max_wrong_over_tie_risk': max(p / max(t, 1) for p, t in zip(per, tie))}
    h.close()
    out[tag] = res
print(json.dumps(out, indent=1))
