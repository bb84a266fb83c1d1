#!/bin/bash
# A/B builds of conv_wino4s_kernel (csrc/wino4s_kernel.hip): NAME:"flags" -> ecseg_amd/libecseg_w4s_NAME.so (every other object is the
# product build's); load with ECSEG_HIP_LIB.  e.g. tools/w4s_variants.sh nofilt:-DECSEG_W4S_ABL=1 nomfma:-DECSEG_W4S_ABL=4
set -e
cd "$(dirname "$0")/../ecseg_amd/csrc"
mkdir -p /tmp/w4s
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize"
for v in "$@"; do n=${v%%:*}; f=${v#*:}; $HC $f -c wino4s_kernel.hip -o /tmp/w4s/w4s_$n.o & done
wait
for v in "$@"; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_w4s_$n.so api.o unet_kernels.o layer_kernels.o wino4_kernel.o /tmp/w4s/w4s_$n.o wino4r_kernel.o convs_kernel.o wino16_kernel.o post_kernels.o host_codec.o host_io.o comm.o -lz -ldl
done
ls -la ../libecseg_w4s_*.so
