"""F(4x4) layers with exactly 32 output channels: the zero-padded 64-channel block (option wino4_split=0) against the split-K
mode (1), for several K lengths and extents.  python tools/w4_split_probe.py (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from ecseg_amd.model import MetasegModel
from tools.layer_probe import cfg_for
rng = np.random.default_rng(0)
for cin, hw, npat in [(64, 64, 280), (128, 64, 280), (256, 64, 280), (64, 128, 70), (64, 256, 35)]:
    w = {'c': [(rng.normal(size=(3, 3, cin, 32)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=32).astype(np.float32)]}
    m = MetasegModel(cfg_for(cin, 32, hw), w)
    x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
    res = []
    for sp in (0, 1):
        m.handle.set_option('wino4_split', sp)
        m.handle.set_kernel_profiling(True)
        m.handle.forward_patches(x); m.handle.conv_profile()
        ms = 0.0
        for _ in range(3):
            m.handle.forward_patches(x); t, nl, fl = m.handle.conv_profile(); ms += t
        res.append(ms / 3)
    print('%d->32 @%d x%d  padded %.3f ms  split %.3f ms' % (cin, hw, npat, res[0], res[1]), flush=True)
    del m
