#!/bin/bash
# A/B builds of conv_wino4r_kernel (csrc/wino4r_kernel.hip): NAME:"flags" -> ecseg_amd/libecseg_w4r_NAME.so; load with ECSEG_HIP_LIB
set -e
cd "$(dirname "$0")/../ecseg_amd/csrc"
mkdir -p /tmp/w4r
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize"
for v in "$@"; do n=${v%%:*}; f=${v#*:}; $HC $f -c wino4r_kernel.hip -o /tmp/w4r/w4r_$n.o & done
wait
for v in "$@"; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_w4r_$n.so api.o unet_kernels.o layer_kernels.o wino4_kernel.o wino4s_kernel.o /tmp/w4r/w4r_$n.o convs_kernel.o wino16_kernel.o post_kernels.o host_codec.o host_io.o comm.o -lz -ldl
done
ls -la ../libecseg_w4r_*.so
