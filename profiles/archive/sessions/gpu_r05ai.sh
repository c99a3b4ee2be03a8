#!/bin/bash
# session r05ai: consecutive tiles per block strided over the channel-call (tiles y, y + gridDim.y, ...)
OUT=gpurun_out/r05ai; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
python tools/ab_libs.py --rounds 3 --cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 8,250000,44100 --cfg 64,37500,8000 --cfg 3,400000,48000 base=$P/libfmd_hip_base.so t1= t2=$P/libfmd_hip_t2.so t3=$P/libfmd_hip_t3.so t6=$P/libfmd_hip_t6.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-200
