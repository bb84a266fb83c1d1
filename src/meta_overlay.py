#!/usr/bin/env python3
# `make meta_overlay` entry point with the reference's path; the implementation is ecseg_amd/meta_overlay.py.
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecseg_amd.meta_overlay import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])
