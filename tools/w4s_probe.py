"""F(4x4) on the bf16 pipe with 3-way split operands (conv_wino4s_kernel, option winograd = 3) against the fp32-MFMA F(4x4) kernel
(winograd = 2) and the direct kernel (0): per layer shape the kernel time and the error against a float64 convolution of the same
float32 inputs (2 patches).  Usage (GPU box): python tools/w4s_probe.py [n_patches] [--quick]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ecseg_amd.model import MetasegModel  # noqa: E402
from tools.layer_probe import cfg_for, SHAPES  # noqa: E402


def ref64(x, w, b):
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1)
    y = torch.nn.functional.conv2d(xt, wt, torch.from_numpy(b.astype(np.float64)), padding=1)
    return torch.relu(y).permute(0, 2, 3, 1).numpy()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if args else 70
    shapes = SHAPES + [(12, 64, 64), (68, 192, 32)]
    if '--quick' in sys.argv:
        shapes = [(64, 64, 256), (256, 256, 64), (1024, 512, 32), (12, 64, 64)]
    rng = np.random.default_rng(0)
    print('%-24s %9s %9s %9s  %6s   %10s %10s %10s' % ('layer', 'direct ms', 'F4x4 ms', 'split ms', 'x', 'err direct', 'err F4x4', 'err split'))
    for cin, cout, hw in shapes:
        npat = n if hw >= 128 else 4 * n
        w = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
        m = MetasegModel(cfg_for(cin, cout, hw), w)
        x = rng.integers(0, 256, size=(npat, hw, hw, cin), dtype=np.uint8)
        want = ref64(x[:2], w['c'][0], w['c'][1])
        scale = max(1.0, float(np.abs(want).max()))
        errs, times = {}, {}
        for mode in (0, 2, 3):
            m.handle.set_option('winograd', mode)
            errs[mode] = float(np.abs(m.handle.forward_patches(x[:2]) - want).max()) / scale
            m.handle.set_kernel_profiling(True)
            m.handle.forward_patches(x)
            m.handle.conv_profile()
            ms = 0.0
            for _ in range(3):
                m.handle.forward_patches(x)
                ms += m.handle.conv_profile()[0]
            times[mode] = ms / 3
            m.handle.set_kernel_profiling(False)
        print('%4d->%4d @%3d x%-6d %9.3f %9.3f %9.3f  %6.2f   %10.2e %10.2e %10.2e' % (
            cin, cout, hw, npat, times[0], times[2], times[3], times[2] / times[3], errs[0], errs[2], errs[3]), flush=True)
        del m


if __name__ == '__main__':
    main()
