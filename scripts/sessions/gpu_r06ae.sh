#!/bin/bash
# session r06ae: the resampler pass's index arithmetic for a lane's first audio sample computed under the loads' latency (pass_index):
# parity / fuzz / guard tests, A/B against the library of commit 4fdefc2 over the rows
OUT=gpurun_out/r06ae; mkdir -p $OUT; export TMPDIR=/tmp
BASE=$PWD/rtl-sdr-rs_amd/libfmd_hip_r06b.so
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_gpu_ref_kat.py -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
timeout 1200 python tools/ab_libs.py --rounds 5 --cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 8,250000,44100 --cfg 7,166666,32000 --cfg 3,400000,48000 --cfg 2,500000,32000 --cfg 12,192000,32000 --cfg 1,48000,48000 --cfg 6,200000,48000 base=$BASE new= 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
