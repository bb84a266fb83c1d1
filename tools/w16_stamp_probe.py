"""In-kernel cycle stamps of conv_wino16_kernel (diagnostic build: `bash tools/build_variants.sh diag`,
`ECSEG_HIP_LIB=.../ecseg_amd/libecseg_diag.so`): per wave, the s_memtime ticks spent in each phase of a stage, for one
workgroup in the middle of the grid."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

SHAPES = [(16, 16, 256, 280), (32, 16, 256, 280), (32, 32, 256, 140)]
NAMES = ['DMA issue', 'reads+T', 'F reads+MFMA', 'DMA wait', 'output', 'barrier']


def main():
    rng = np.random.default_rng(0)
    for cin, cout, hw, npat in SHAPES:
        w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                   rng.normal(size=cout).astype(np.float32)]}
        m = MetasegModel(cfg_for(cin, cout, hw), w)
        x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
        m.handle.set_option('winograd', 1)
        m.handle.forward_patches(x)
        m.handle.forward_patches(x)
        d = m.handle.debug_peek(64).reshape(8, 8)
        ns = d[0, 7]
        print('%d->%d@%d  stages %d; ticks per stage:' % (cin, cout, hw, ns))
        print('  wave ' + ' '.join('%13s' % n for n in NAMES) + '      total/stage')
        for wv in range(8):
            print('  %4d ' % wv + ' '.join('%13.1f' % (d[wv, i] / ns) for i in range(6)) + '   %10.1f' % (d[wv, 6] / ns))
        del m


if __name__ == '__main__':
    main()
