#!/bin/bash
# Side builds of the library for A/B timing through ECSEG_HIP_LIB (other objects are reused from the product build):
#   diag      ecseg_amd/libecseg_diag.so with -DECSEG_DIAG: the timing-only ablation kernels (ECSEG_W4_ABL) and the in-kernel
#             cycle stamps (tools/w4_stamp_probe.py) exist only there (csrc/wino4_diag.inc), never in the shipped library
#   points12  ecseg_amd/libecseg_points12.so: the F(4x4) kernel with the textbook interpolation points {0, +-1, +-2, inf}
#   <flags>   anything else is passed as -D<flags> to wino4_kernel.hip -> ecseg_amd/libecseg_v<flags>.so
set -e
cd "$(dirname "$0")/../ecseg_amd/csrc"
mkdir -p /tmp/w4
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC"
for v in "$@"; do
  if [ "$v" = diag ]; then
    $HC -fno-slp-vectorize -DECSEG_DIAG -c wino4_kernel.hip -o /tmp/w4/wino4_diag.o &
    $HC -DECSEG_DIAG -c api.hip -o /tmp/w4/api_diag.o &
    $HC -DECSEG_DIAG -c wino16_kernel.hip -o /tmp/w4/wino16_diag.o &
  elif [ "$v" = points12 ]; then
    $HC -fno-slp-vectorize -DECSEG_W4_PA=1 -DECSEG_W4_PB=2 -c wino4_kernel.hip -o /tmp/w4/wino4_p12.o &
    $HC -DECSEG_W4_PA=1 -DECSEG_W4_PB=2 -c api.hip -o /tmp/w4/api_p12.o &
  else
    $HC -fno-slp-vectorize -D$v -c wino4_kernel.hip -o /tmp/w4/wino4_v$v.o &
  fi
done
wait
for v in "$@"; do
  if [ "$v" = diag ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_diag.so /tmp/w4/api_diag.o unet_kernels.o /tmp/w4/wino4_diag.o /tmp/w4/wino16_diag.o post_kernels.o host_codec.o host_io.o comm.o -lz -ldl
  elif [ "$v" = points12 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_points12.so /tmp/w4/api_p12.o unet_kernels.o /tmp/w4/wino4_p12.o wino16_kernel.o post_kernels.o host_codec.o host_io.o comm.o -lz -ldl
  else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libecseg_v$v.so api.o unet_kernels.o /tmp/w4/wino4_v$v.o wino16_kernel.o post_kernels.o host_codec.o host_io.o comm.o -lz -ldl
  fi
done
ls -la ../libecseg_*.so
