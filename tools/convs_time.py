"""Kernel time of the decoder's 2x2 / stride-2 up-convolutions under winograd = 2 (conv_mfma_kernel, fp32 MFMA) and 3 (convs_kernel, bf16x3)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402

rng = np.random.default_rng(0)
out = []
for cin, cout, hw in [(1024, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, hw, hw, cin]}, 'inbound_nodes': []},
        {'class_name': 'Conv2DTranspose', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [2, 2], 'strides': [2, 2],
                                                                  'padding': 'same', 'activation': 'relu', 'use_bias': True},
         'inbound_nodes': [[['in', 0, 0, {}]]]}], 'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    w = {'c': [(rng.normal(size=(2, 2, cout, cin)) / np.sqrt(cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    m = MetasegModel(cfg, w)
    npat = 280 if hw <= 64 else 70
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
    out.append('%d->%d@%d x%d %.3f/%.3f' % (cin, cout, hw, npat, t[2], t[3]))
    del m
print(' | '.join(out), flush=True)
