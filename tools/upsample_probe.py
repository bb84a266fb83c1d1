"""UpSampling2D + Conv2D(2x2) decoder: device U-Net time of the canonical `upsample` models with the round-6 lowering (one 3x3 / stride-2
transposed convolution per decoder step) and with the lowering switched off (a real nearest up-sampling pass + a 2x2 convolution, what rounds
1 - 5 ran).  python tools/upsample_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd import keras_plan, synth  # noqa: E402
from ecseg_amd._lib import Handle  # noqa: E402

orig = keras_plan.build_plan
for base, nimg in ((64, 16), (16, 64)):
    cfg = synth.unet_config(base=base, up='upsample')
    w = synth.unet_weights(cfg, seed=0)
    imgs = np.stack([synth.dapi_image(i) for i in range(nimg)])
    out = {}
    for lowered in (True, False):
        h = Handle(0)
        plan = orig(cfg, w)
        if not lowered:
            # the plan of rounds 1 - 5: peephole fusion of BatchNorm / activations only
            import unittest.mock as mock
            src = open(keras_plan.__file__).read()
            with mock.patch.dict(os.environ, {}):
                ns = {}
                code = src.replace("if u['kind'] != 'upsample' or", "if True or u['kind'] != 'upsample' or")
                mod = type(keras_plan)('keras_plan_nolower')
                mod.__file__ = keras_plan.__file__
                exec(compile(code, keras_plan.__file__, 'exec'), mod.__dict__)
                plan = mod.build_plan(cfg, w)
        h.load_plan(plan)
        h.segment_images(imgs)
        ms = []
        for _ in range(3):
            h.segment_images(imgs)
            ms.append(h.timings()['unet'])
        out[lowered] = (min(ms) / nimg, sum(1 for o in plan.ops if o['op'] == keras_plan.OP_UPSAMPLE))
        h.close()
    print('base %d, %d images: U-Net %.3f ms/image lowered (%d UpSampling2D ops) | %.3f ms/image as written (%d) | x%.2f' % (
        base, nimg, out[True][0], out[True][1], out[False][0], out[False][1], out[False][0] / out[True][0]), flush=True)
