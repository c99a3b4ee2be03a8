#!/bin/bash
# session r05ad: downsample 3 (rates 334 k ... 500 k of optimal_settings) -- where it stands, and its tiling
OUT=gpurun_out/r05ad; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 3 --cfg 3,400000,48000 --cfg 3,500000,32000 shipped= exp=$X kt256=$X@FMD_KT=256 kt384=$X@FMD_KT=384 kt448=$X@FMD_KT=448 kt512=$X@FMD_KT=512 kt640=$X@FMD_KT=640 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-250
