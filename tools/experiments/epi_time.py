"""Kernel time of conv_wino4_kernel (wino4_rowpass = 0) on three short-K layer shapes: the A/B harness of the output-stage
ablation builds (git apply -p0 tools/experiments/w4_epilogue_ablation_hooks.patch, then tools/build_variants.sh ECSEG_W4_EPI_ABL=<bits>: 1 no output stores, 2 no LDS reads in the combine step, 4 no
combine step, 8 no fold + write of the exchange image; results are wrong on purpose)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

rng = np.random.default_rng(0)
out = []
for cin, cout, hw, npat in [(64, 64, 256, 70), (128, 128, 128, 70), (256, 256, 64, 280)]:
    w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    m = MetasegModel(cfg_for(cin, cout, hw), w)
    x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
    m.handle.set_option('wino4_rowpass', int(os.environ.get('ROWPASS', '0')))
    m.handle.set_kernel_profiling(True)
    m.handle.forward_patches(x)
    m.handle.conv_profile()
    ms = []
    for _ in range(5):
        m.handle.forward_patches(x)
        ms.append(m.handle.conv_profile()[0])
    out.append('%d->%d@%d %.3f' % (cin, cout, hw, min(ms)))
    del m
print(os.path.basename(os.environ.get('ECSEG_HIP_LIB', 'libecseg_hip.so')), ' | '.join(out), flush=True)
