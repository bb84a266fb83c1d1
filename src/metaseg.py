#!/usr/bin/env python3
# `make metaseg` entry point with the reference's path (src/metaseg.py); the implementation is ecseg_amd/metaseg.py.
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecseg_amd.metaseg import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])
