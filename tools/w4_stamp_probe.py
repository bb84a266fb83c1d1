"""In-kernel cycle stamps of conv_wino4_kernel (run with ECSEG_W4_ABL=100 on the diagnostic build: `bash
tools/build_variants.sh diag`, `ECSEG_HIP_LIB=.../ecseg_amd/libecseg_diag.so`): per wave and 8-channel group, the cycles
spent in each phase of the main loop, for one workgroup in the middle of the grid."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for  # noqa: E402

SHAPES = [(64, 64, 256, 70), (512, 256, 64, 280), (1024, 1024, 16, 280)]
NAMES = ['vmwait', 'barrier', 'transform', 'mfma0+dma', 'filt wait', 'mfma1+dma', 'haloDMA', '-']


def main():
    rng = np.random.default_rng(0)
    for cin, cout, hw, npat in SHAPES:
        w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                   rng.normal(size=cout).astype(np.float32)]}
        m = MetasegModel(cfg_for(cin, cout, hw), w)
        x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
        m.handle.set_option('winograd', 2)
        m.handle.set_option('wino4_rowpass', 0)          # the stamps live in conv_wino4_kernel (same output stage as conv_wino4r_kernel)
        m.handle.forward_patches(x)
        m.handle.forward_patches(x)
        raw = m.handle.debug_peek(168)
        d = raw[:120].reshape(12, 10)
        ep = raw[120:168].reshape(12, 4)
        ng = d[0, 9]
        print('%d->%d@%d  groups %d; cycles per group (s_memtime ticks x ~24 at 100 MHz? raw ticks shown):' % (cin, cout, hw, ng))
        print('  wave ' + ' '.join('%10s' % n for n in NAMES) + '      total/grp')
        for wv in range(12):
            print('  %4d ' % wv + ' '.join('%10.1f' % (d[wv, i] / ng) for i in range(6)) + '   %10.1f' % (d[wv, 8] / ng) +
                  '   | prologue %7.0f  K loop %8.0f  whole WG %8.0f  (epilogue %7.0f)' % (d[wv, 6], d[wv, 8], d[wv, 7], d[wv, 7] - d[wv, 8] - d[wv, 6]))
        print('  output stage per wave (both passes): barrier / fold+write R / barrier / combine+stores')
        for wv in range(12):
            print('  %4d ' % wv + ' '.join('%9.0f' % v for v in ep[wv]))
        del m


if __name__ == '__main__':
    main()
