#!/bin/bash
# session r05bj: shipped library, 8-bit against 12-bit taps at rates with an odd column parameter (8-bit: the even one below, one sparse digit)
OUT=gpurun_out/r05bj; mkdir -p $OUT; export TMPDIR=/tmp
for rates in "1000000 44100" "1400000 48000" "2500000 48000"; do set -- $rates
python tools/ab_libs.py --firdemod --fd-fast $1 --fd-slow $2 --rounds 4 shipped= 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --fd-fast $1 --fd-slow $2 --fir-taps-max 127 --rounds 4 shipped= 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
done
