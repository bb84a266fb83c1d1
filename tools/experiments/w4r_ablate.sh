# A/B of conv_wino4r_kernel builds (tools/w4r_variants.sh): kernel time per layer shape (wino4 / wino4r columns of tools/w4r_time.py)
python3 tools/w4r_time.py 70
for v in "$@"; do echo "== $v"; ECSEG_HIP_LIB=$PWD/ecseg_amd/libecseg_w4r_$v.so python3 tools/w4r_time.py 70; done
