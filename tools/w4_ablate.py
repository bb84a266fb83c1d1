"""Timing of conv_wino4_kernel on a few layer shapes; run once per ECSEG_W4_ABL value (timing-only ablations).
Needs the diagnostic build: `bash tools/build_variants.sh diag` and `ECSEG_HIP_LIB=.../ecseg_amd/libecseg_diag.so` (the
shipped library contains no ablation kernels and ignores ECSEG_W4_ABL)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

SHAPES = [(64, 64, 256, 70), (256, 128, 128, 70), (512, 256, 64, 280), (1024, 512, 32, 280), (1024, 1024, 16, 280)]


def main():
    rng = np.random.default_rng(0)
    out = []
    for cin, cout, hw, npat in SHAPES:
        w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                   rng.normal(size=cout).astype(np.float32)]}
        m = MetasegModel(cfg_for(cin, cout, hw), w)
        x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
        m.handle.set_option('winograd', 2)
        m.handle.forward_patches(x)
        m.handle.set_kernel_profiling(True)
        ms = 0.0
        for _ in range(3):
            m.handle.forward_patches(x)
            ms += m.handle.conv_profile()[0]
        ms /= 3
        fl = 2.0 * 9 * cin * cout * hw * hw * npat
        out.append('%d->%d@%d %.3f ms (%.0f TF alg, %.0f%% mfma)' % (cin, cout, hw, ms, fl / ms / 1e9, fl / 4 / ms / 1e9 / 157.3 * 100))
        del m
    print('ABL=%s  ' % os.environ.get('ECSEG_W4_ABL', '0') + ' | '.join(out), flush=True)


if __name__ == '__main__':
    main()
