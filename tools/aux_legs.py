#!/usr/bin/env python3
"""The overlay / meta_preprocess / all-gather legs of bench.py (aux_device_legs) on their own - the program to put behind
`rocprofv3 --kernel-trace --stats --` for the kernel summary of BASELINE configs[4] (tools/aux_profile.sh).
    python3 tools/aux_legs.py [--images 64]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=64)
    ap.add_argument('--no-comm', action='store_true', help='skip the one-rank all-gather (under rocprofv3)')
    a = ap.parse_args()
    import bench
    from ecseg_amd._lib import Handle
    h = Handle(0)
    print(json.dumps(bench.aux_device_legs(h, 0, a.images, with_comm=not a.no_comm)))
    h.close()


if __name__ == '__main__':
    main()
