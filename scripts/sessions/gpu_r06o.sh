#!/bin/bash
# session r06o: the single-point form of the i32 wrap at downsample 4 (two instructions instead of four per sample): parity (the new
# diagonal full-scale test, the whole parity / fuzz files), then same-process A/B against the library of commit 4ae6769
OUT=gpurun_out/r06o; mkdir -p $OUT; export TMPDIR=/tmp
BASE=$PWD/rtl-sdr-rs_amd/libfmd_hip_r06base.so
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
FMD_FUZZ_CASES=600 FMD_FUZZ_SEED=60603 timeout 600 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu >> $OUT/tests.log 2>&1; echo "fuzz rc=$?" >> $OUT/tests.log
timeout 900 python tools/ab_libs.py --rounds 5 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 4,250000,44100 --cfg 2,500000,32000 --cfg ref base=$BASE new= 2>/dev/null | tee $OUT/ab_wrap1.txt | cut -c1-260
