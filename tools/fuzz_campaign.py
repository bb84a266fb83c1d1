#!/usr/bin/env python3
"""Long randomised parity campaign (device vs CPU oracle, bit-exact) beyond the fixed seeds of tests/test_gpu_fuzz.py:
larger images (many tiles, multi-tile giant components), more seeds.  Runs for --seconds, prints one line per failure and a
summary; exit code 1 on any mismatch.

    python tools/fuzz_campaign.py --seconds 300 [--seed0 0]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=300)
    ap.add_argument('--seed0', type=int, default=0)
    ap.add_argument('--seeds', default=None, help='comma-separated list: run exactly these seeds')
    a = ap.parse_args()
    import test_gpu_fuzz as F
    from ecseg_amd._lib import Handle
    from oracle import postproc
    gpu = Handle(0)
    t0 = time.time()
    seed = a.seed0
    todo = [int(x) for x in a.seeds.split(',')] if a.seeds else None
    n_ccl = n_meta = fails = 0
    while time.time() - t0 < a.seconds:
        if todo is not None:
            if not todo:
                break
            seed = todo.pop(0)
        m, labs = F._campaign_case(seed)
        n, H, W = m.shape
        for conn, fn in ((8, postproc.label8), (4, postproc.label4)):
            got = gpu.ccl_labels(m, conn)
            for k in range(n):
                if not np.array_equal(got[k], F._canon(fn(m[k])[0])):
                    print('FAIL ccl seed %d conn %d %dx%d image %d' % (seed, conn, H, W, k), flush=True)
                    fails += 1
        cnt, px = gpu.count_cc(m)
        for k in range(n):
            wn, wpx = postproc.count_cc(m[k].astype(bool))
            if int(cnt[k]) != wn or not (int(px[k]) == wpx or (px[k] == -1 and wpx == 0.0)):
                print('FAIL count_cc seed %d %dx%d image %d' % (seed, H, W, k), flush=True)
                fails += 1
        n_ccl += n
        if labs is not None:
            out, nec = gpu.meta_inference(labs)
            for k in range(n):
                want = postproc.meta_inference(labs[k])
                if not np.array_equal(out[k], want) or int(nec[k]) != postproc.count_cc(want == 3)[0]:
                    print('FAIL meta_inference seed %d %dx%d image %d (%d px differ)' % (seed, H, W, k, int((out[k] != want).sum())), flush=True)
                    fails += 1
            n_meta += n
        seed += 1
    print('fuzz campaign: seeds %d..%d, %d label maps through both labellings + count_cc, %d through meta_inference, %d failure(s), %.0f s'
          % (a.seed0, seed - 1, n_ccl, n_meta, fails, time.time() - t0), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
