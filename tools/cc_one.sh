#!/bin/bash
# Compile ONE kernel source of ecseg_amd/csrc with the library's flags, keep the ISA, print registers / spills per kernel.
#   tools/cc_one.sh wino4s_kernel.hip [extra hipcc flags]
set -e
src=$1; shift
out=/root/repo/build/cc_one; mkdir -p $out
cd /root/repo/ecseg_amd/csrc
extra=""
case $src in wino4*_kernel.hip) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function $extra "$@" -save-temps=obj -c $src -o $out/${src%.hip}.o 2>&1 | grep -v "argument unused" || true
python3 - $out/${src%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for blk in s.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    print('%-84s vgpr %s spill %s sgpr %s scratch %s' % (g('name')[:84], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('private_segment_fixed_size')))

PY
