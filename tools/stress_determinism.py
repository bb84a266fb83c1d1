"""Race screen: the device pipeline must return bit-identical labels / counts on every repetition (base-64 model,
full-size images); also repeats single F(4x4) layers and compares the float outputs bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd import synth  # noqa: E402
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    cfg = synth.unet_config(base=64)
    m = MetasegModel(cfg, synth.unet_weights(cfg, seed=0))
    imgs = np.stack([synth.dapi_image(i, 1040, 1392) for i in range(4)])
    ref = m.handle.segment_images(imgs, want_raw=True)
    bad = 0
    for r in range(reps):
        out = m.handle.segment_images(imgs, want_raw=True)
        for a, b in zip(ref, out):
            if not np.array_equal(a, b):
                bad += 1
    print('pipeline: %d repetitions, %d mismatching outputs' % (reps, bad))
    del m
    rng = np.random.default_rng(1)
    for cin, cout, hw, n in [(64, 64, 256, 8), (512, 256, 64, 64), (1024, 1024, 16, 256)]:
        w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
        mm = MetasegModel(cfg_for(cin, cout, hw), w)
        x = rng.integers(0, 256, size=(n, hw, hw, cin), dtype=np.uint8)
        for mode, rowpass in ((2, 1), (2, 0), (3, 1)):       # conv_wino4r_kernel (default), conv_wino4_kernel, conv_wino4s_kernel (bf16x3 split)
            mm.handle.set_option('winograd', mode)
            mm.handle.set_option('wino4_rowpass', rowpass)
            r0 = mm.handle.forward_patches(x)
            nb = sum(not np.array_equal(r0, mm.handle.forward_patches(x)) for _ in range(reps))
            print('layer %d->%d@%d winograd=%d rowpass=%d: %d repetitions, %d mismatching' % (cin, cout, hw, mode, rowpass, reps, nb))
            bad += nb
        del mm
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
