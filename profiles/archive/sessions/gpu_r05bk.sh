#!/bin/bash
# session r05bk: rounds in flight per wave of the register-streaming kernel (downsample 2 / 4): 3 / 4 (shipped) / 5 / 6 / 8, shipped-flavour builds side by side
OUT=gpurun_out/r05bk; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 2,500000,32000 --cfg 2,96000,48000 p4= p3=$P/libfmd_hip_p3.so p5=$P/libfmd_hip_p5.so p6=$P/libfmd_hip_p6.so p8=$P/libfmd_hip_p8.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-200
