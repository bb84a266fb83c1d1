# A/B of conv_wino4s_kernel builds (tools/w4s_variants.sh): kernel time per layer shape, fp32 F(4x4) / split
set -e
python3 tools/w4s_time.py 70
for v in "$@"; do ECSEG_HIP_LIB=$PWD/ecseg_amd/libecseg_w4s_$v.so python3 tools/w4s_time.py 70; done
