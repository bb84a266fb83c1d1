"""Kernel time of a few 3x3 layer shapes under winograd = 2 and 3 for the library named by ECSEG_HIP_LIB (A/B of tools/w4s_variants.sh
builds; results are not checked).  python tools/w4s_time.py [n_patches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
rng = np.random.default_rng(0)
out = []
for cin, cout, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 1024, 16)]:
    npat = n if hw >= 128 else 4 * n
    w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    m = MetasegModel(cfg_for(cin, cout, hw), w)
    x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
    t = {}
    for mode in (2, 3):
        m.handle.set_option('winograd', mode)
        m.handle.set_kernel_profiling(True)
        m.handle.forward_patches(x)
        m.handle.conv_profile()
        ms = 0.0
        for _ in range(3):
            m.handle.forward_patches(x)
            ms += m.handle.conv_profile()[0]
        t[mode] = ms / 3
        m.handle.set_kernel_profiling(False)
    out.append('%d->%d@%d %.3f/%.3f' % (cin, cout, hw, t[2], t[3]))
    del m
print(os.path.basename(os.environ.get('ECSEG_HIP_LIB', 'product')), ' | '.join(out), flush=True)
