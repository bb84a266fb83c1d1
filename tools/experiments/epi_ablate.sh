#!/bin/bash
# Output-stage ablations of conv_wino4_kernel, one box: git apply -p0 tools/experiments/w4_epilogue_ablation_hooks.patch && bash tools/build_variants.sh ECSEG_W4_EPI_ABL=1 ... first.
for lib in hip vECSEG_W4_EPI_ABL=1 vECSEG_W4_EPI_ABL=2 vECSEG_W4_EPI_ABL=3 vECSEG_W4_EPI_ABL=4 vECSEG_W4_EPI_ABL=12 hip; do
  ECSEG_HIP_LIB=$PWD/ecseg_amd/libecseg_$lib.so timeout -k 10 200 python3 tools/experiments/epi_time.py 2>/dev/null | tail -1
done
