#!/bin/bash
# session r05bh: 8-bit filters whose audio groups admit an odd column parameter: the even one below it with one sparse digit, against two dense digits at the odd one
OUT=gpurun_out/r05bh; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 1200 python -m pytest tests/test_firdemod.py -x -q -m gpu -k "one_digit or variants" 2>&1 | tail -3 | tee $OUT/pytest.log
for rates in "1000000 44100" "1400000 48000"; do set -- $rates
python tools/ab_libs.py --firdemod --fd-fast $1 --fd-slow $2 --fir-taps-max 127 --rounds 4 even_one_digit=$X odd_two_digits=$X@FMD_FD_DIGITS=2 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
done
