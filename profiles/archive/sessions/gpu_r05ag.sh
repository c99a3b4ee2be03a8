#!/bin/bash
# session r05ag: consecutive tiles per block, shipped-flavour builds side by side (base = the committed one-tile kernels)
OUT=gpurun_out/r05ag; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
python tools/ab_libs.py --rounds 4 --cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 8,250000,44100 --cfg 7,166666,32000 --cfg 64,37500,8000 --cfg 3,400000,48000 --cfg 1,48000,48000 --cfg 12,192000,32000 --cfg 16,150000,32000 base=$P/libfmd_hip_base.so t1= t2=$P/libfmd_hip_t2.so t3=$P/libfmd_hip_t3.so t4=$P/libfmd_hip_t4.so t6=$P/libfmd_hip_t6.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-200
