#!/bin/bash
# session r06ac: the resampler pass's kernel-argument reads in one scalar round trip in front of the block barrier (three dependent ones
# behind it before): parity / fuzz, A/B against the library of commit 4fdefc2 over the rows
OUT=gpurun_out/r06ac; mkdir -p $OUT; export TMPDIR=/tmp
BASE=$PWD/rtl-sdr-rs_amd/libfmd_hip_r06b.so
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
timeout 1200 python tools/ab_libs.py --rounds 5 --cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 8,250000,44100 --cfg 7,166666,32000 --cfg 3,400000,48000 --cfg 2,500000,32000 --cfg 12,192000,32000 --cfg 1,48000,48000 base=$BASE new= 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
