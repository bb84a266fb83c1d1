#!/bin/bash
# Build libecseg_v<N>.so for each W4_VARIANT given (A/B timing through ECSEG_HIP_LIB); other objects are reused.
set -e
cd "$(dirname "$0")/../ecseg_amd/csrc"
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DW4_VARIANT=$v -c wino4_kernel.hip -o /tmp/w4/wino4_v$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_v$v.so api.o unet_kernels.o /tmp/w4/wino4_v$v.o post_kernels.o host_codec.o
done
ls -la ../libecseg_v*.so
