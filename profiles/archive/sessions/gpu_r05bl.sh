#!/bin/bash
# session r05bl: 12-bit filters at an odd column parameter (dense two digits) against the even one below it (sparse two digits), shipped-flavour builds
OUT=gpurun_out/r05bl; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
for rates in "1000000 44100" "1400000 48000" "1000000 48000" "1500000 48000"; do set -- $rates
python tools/ab_libs.py --firdemod --fd-fast $1 --fd-slow $2 --rounds 4 odd_dense= even_sparse=$P/libfmd_hip_ev.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
done
