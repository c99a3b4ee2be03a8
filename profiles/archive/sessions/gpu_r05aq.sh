#!/bin/bash
# session r05aq: tile size of the register-streaming kernel (downsample 4 and 2) on the final kernels
OUT=gpurun_out/r05aq; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 3 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 2,500000,32000 shipped= exp=$X k256=$X@FMD_KT_STREAM=256 k384=$X@FMD_KT_STREAM=384 k512=$X@FMD_KT_STREAM=512 k768=$X@FMD_KT_STREAM=768 k1536=$X@FMD_KT_STREAM=1536 k2048=$X@FMD_KT_STREAM=2048 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
