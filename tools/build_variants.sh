#!/bin/bash
# Side builds of the library for A/B timing through ECSEG_HIP_LIB (other objects are reused from the product build):
#   diag            ecseg_amd/libecseg_diag.so with -DECSEG_DIAG: the timing-only ablation kernels (ECSEG_W4_ABL) and the
#                   in-kernel cycle stamps (tools/w4_stamp_probe.py) exist only there (csrc/wino4_diag.inc)
#   points12        ecseg_amd/libecseg_points12.so: the F(4x4) kernel with the textbook interpolation points {0, +-1, +-2, inf}
#   <name>:<flags>  wino4_kernel.hip AND api.hip with <flags> (e.g. "freg:-DECSEG_W4_FREG=1 -DECSEG_W4_TSLOTS=4": the filter
#                   image layout is api.hip's) -> ecseg_amd/libecseg_v<name>.so
#   <flag>          anything else is passed as -D<flag> to wino4_kernel.hip only -> ecseg_amd/libecseg_v<flag>.so
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
  elif [[ "$v" == *:* ]]; then
    n=${v%%:*}; f=${v#*:}
    $HC -fno-slp-vectorize $f -c wino4_kernel.hip -o /tmp/w4/wino4_v$n.o &
    $HC $f -c api.hip -o /tmp/w4/api_v$n.o &
  else
    $HC -fno-slp-vectorize -D$v -c wino4_kernel.hip -o /tmp/w4/wino4_v$v.o &
  fi
done
wait
LK="/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC"
REST="layer_kernels.o wino4s_kernel.o wino4r_kernel.o convs_kernel.o post_kernels.o host_codec.o host_io.o comm.o -lz -ldl"   # objects of the product build
for v in "$@"; do
  if [ "$v" = diag ]; then
    $LK -o ../libecseg_diag.so /tmp/w4/api_diag.o unet_kernels.o /tmp/w4/wino4_diag.o /tmp/w4/wino16_diag.o $REST
  elif [ "$v" = points12 ]; then
    $LK -o ../libecseg_points12.so /tmp/w4/api_p12.o unet_kernels.o /tmp/w4/wino4_p12.o wino16_kernel.o $REST
  elif [[ "$v" == *:* ]]; then
    n=${v%%:*}
    $LK -o ../libecseg_v$n.so /tmp/w4/api_v$n.o unet_kernels.o /tmp/w4/wino4_v$n.o wino16_kernel.o $REST
  else
    $LK -o ../libecseg_v$v.so api.o unet_kernels.o /tmp/w4/wino4_v$v.o wino16_kernel.o $REST
  fi
done
ls -la ../libecseg_*.so
