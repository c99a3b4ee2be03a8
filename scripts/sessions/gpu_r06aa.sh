#!/bin/bash
# session r06aa: the streaming kernels' resampler pass with two audio samples per lane and trip: parity, A/B against the library of 4fdefc2
OUT=gpurun_out/r06aa; mkdir -p $OUT; export TMPDIR=/tmp
BASE=$PWD/rtl-sdr-rs_amd/libfmd_hip_r06b.so
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
timeout 900 python tools/ab_libs.py --rounds 5 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 4,250000,44100 --cfg 2,500000,32000 --cfg 2,500000,48000 --cfg ref base=$BASE new= 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
