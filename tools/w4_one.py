"""One conv_wino4 layer shape, a few launches (workload for rocprofv3 --pmc runs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

cin, cout, hw, npat = 512, 256, 64, 280
rng = np.random.default_rng(0)
w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
m = MetasegModel(cfg_for(cin, cout, hw), w)
x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
m.handle.set_option('winograd', int(sys.argv[1]) if len(sys.argv) > 1 else 2)
for _ in range(3):
    m.handle.forward_patches(x)
print('done')
