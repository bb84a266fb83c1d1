"""conv_wino4_kernel against conv_wino4r_kernel (option wino4_rowpass): kernel time per layer shape + max difference of the results."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
rng = np.random.default_rng(0)
for cin, cout, hw in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 512, 32), (1024, 1024, 16), (12, 64, 64)]:
    npat = n if hw >= 128 else 4 * n
    w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    m = MetasegModel(cfg_for(cin, cout, hw), w)
    x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
    t, outs = {}, {}
    for rp in (0, 1):
        m.handle.set_option('wino4_rowpass', rp)
        outs[rp] = m.handle.forward_patches(x[:2])
        m.handle.set_kernel_profiling(True)
        m.handle.forward_patches(x)
        m.handle.conv_profile()
        ms = 0.0
        for _ in range(3):
            m.handle.forward_patches(x)
            ms += m.handle.conv_profile()[0]
        t[rp] = ms / 3
        m.handle.set_kernel_profiling(False)
    print('%4d->%4d @%3d x%-5d  wino4 %.3f ms  wino4r %.3f ms  x%.3f   max |diff| %.2e (scale %.1f)' % (
        cin, cout, hw, npat, t[0], t[1], t[0] / t[1], float(np.abs(outs[0] - outs[1]).max()), float(np.abs(outs[0]).max())), flush=True)
    del m
