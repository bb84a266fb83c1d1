"""Per-layer timing of one 3x3 convolution shape under the three kernels (direct, F(2x2), F(4x4)) + max error vs the
direct kernel.  Usage (GPU box): python tools/layer_probe.py [n_patches]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402

SHAPES = [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64),
          (512, 256, 64), (256, 512, 32), (512, 512, 32), (1024, 512, 32), (512, 1024, 16), (1024, 1024, 16)]


def cfg_for(cin, cout, hw):
    return {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, hw, hw, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    rng = np.random.default_rng(0)
    print('%-22s %10s %10s %10s   TF(alg) d/w2/w4        err w2     err w4' % ('layer', 'direct ms', 'F2x2 ms', 'F4x4 ms'))
    for cin, cout, hw in SHAPES:
        npat = n if hw >= 128 else 4 * n
        w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                   rng.normal(size=cout).astype(np.float32)]}
        m = MetasegModel(cfg_for(cin, cout, hw), w)
        x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
        outs, times = {}, {}
        for mode in (0, 1, 2):
            m.handle.set_option('winograd', mode)
            m.handle.set_kernel_profiling(True)
            outs[mode] = m.handle.forward_patches(x)
            m.handle.conv_profile()
            ms = 0.0
            for _ in range(3):
                m.handle.forward_patches(x[:npat])
                t, nl, fl = m.handle.conv_profile()
                ms += t
            times[mode] = ms / 3
            m.handle.set_kernel_profiling(False)
        fl = 2.0 * 9 * cin * cout * hw * hw * npat
        scale = max(1.0, float(np.abs(outs[0]).max()))
        print('%4d->%4d @%3d x%-5d %10.3f %10.3f %10.3f   %6.1f %6.1f %6.1f   %9.2e %9.2e' % (
            cin, cout, hw, npat, times[0], times[1], times[2], fl / times[0] / 1e9, fl / times[1] / 1e9, fl / times[2] / 1e9,
            np.abs(outs[1] - outs[0]).max() / scale, np.abs(outs[2] - outs[0]).max() / scale), flush=True)
        del m


if __name__ == '__main__':
    main()
