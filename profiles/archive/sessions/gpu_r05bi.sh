#!/bin/bash
# session r05bi: 12-bit taps at rates whose audio groups admit an odd column parameter (1 M -> 44.1 k: 5; 1.4 M -> 48 k: 7): what runs, how fast, and the even one below
OUT=gpurun_out/r05bi; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
for rates in "1000000 44100" "1400000 48000" "1000000 48000"; do set -- $rates
python tools/ab_libs.py --firdemod --fd-fast $1 --fd-slow $2 --rounds 3 shipped= exp=$X ng4=$X@FMD_FD_REG=4 ng5=$X@FMD_FD_REG=5 ng6=$X@FMD_FD_REG=6 ng7=$X@FMD_FD_REG=7 off=$X@FMD_FD_REG=0 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
done
