#!/bin/bash
# session r06v: the streaming kernel at downsample 4 with one-shot waves (8 rounds, every load up front) and the single-point wrap, as
# shipped: the whole GPU suite, a long fuzz run, then same-process A/B against the library of commit 4ae6769
OUT=gpurun_out/r06v; mkdir -p $OUT; export TMPDIR=/tmp
BASE=$PWD/rtl-sdr-rs_amd/libfmd_hip_r06base.so
timeout 1500 python3 -m pytest tests -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
FMD_FUZZ_CASES=800 FMD_FUZZ_SEED=60604 timeout 600 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu >> $OUT/tests.log 2>&1; echo "fuzz rc=$?" >> $OUT/tests.log
timeout 900 python tools/ab_libs.py --rounds 5 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 4,250000,44100 --cfg 4,333333,48000 --cfg 2,500000,32000 --cfg ref --cfg 24 base=$BASE new= 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
