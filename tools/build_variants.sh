#!/bin/bash
# Build ecseg_amd/libecseg_v<N>.so for each W4_VARIANT given (A/B timing through ECSEG_HIP_LIB); other objects are reused.
# "diag" builds libecseg_diag.so with -DECSEG_DIAG: the timing-only ablation kernels (ECSEG_W4_ABL) and the in-kernel
# cycle stamps (tools/w4_stamp_probe.py) exist only there, never in the shipped library.
set -e
cd "$(dirname "$0")/../ecseg_amd/csrc"
mkdir -p /tmp/w4
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC"
for v in "$@"; do
  if [ "$v" = diag ]; then
    $HC -fno-slp-vectorize -DECSEG_DIAG -c wino4_kernel.hip -o /tmp/w4/wino4_diag.o &
    $HC -DECSEG_DIAG -c api.hip -o /tmp/w4/api_diag.o &
    $HC -DECSEG_DIAG -c wino16_kernel.hip -o /tmp/w4/wino16_diag.o &
  else
    $HC -fno-slp-vectorize -DW4_VARIANT=$v -c wino4_kernel.hip -o /tmp/w4/wino4_v$v.o &
  fi
done
wait
for v in "$@"; do
  if [ "$v" = diag ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_diag.so /tmp/w4/api_diag.o unet_kernels.o /tmp/w4/wino4_diag.o /tmp/w4/wino16_diag.o post_kernels.o host_codec.o
  else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_v$v.so api.o unet_kernels.o /tmp/w4/wino4_v$v.o wino16_kernel.o post_kernels.o host_codec.o
  fi
done
ls -la ../libecseg_*.so
