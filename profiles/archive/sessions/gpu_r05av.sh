#!/bin/bash
# session r05av: the one-digit form of the stand-alone FIR kernel: its tests, then 8-bit taps through one and two digits side by side
OUT=gpurun_out/r05av; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fir.py tests/test_gpu_ref_kat.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.log
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --fir --fir-taps-max 127 --rounds 4 shipped= one=$X two=$X@FMD_FIR_DIGITS=2 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --fir --fir-taps-max 127 --out-bufs 4 --rounds 3 one=$X two=$X@FMD_FIR_DIGITS=2 2>/dev/null | sed 's/^/{"out_bufs": 4, /; s/, {/, /' | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --fir --fir-taps-max 127 --rounds 3 one_nostores=$X@FMD_DBG=4 two_nostores=$X@FMD_DBG=4,FMD_FIR_DIGITS=2 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
