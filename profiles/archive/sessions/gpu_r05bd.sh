#!/bin/bash
# session r05bd: the stand-alone FIR's one-digit form with its two outputs in one 16-byte store: tests, one against two digits again
OUT=gpurun_out/r05bd; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fir.py tests/test_gpu_ref_kat.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.log
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --fir --fir-taps-max 127 --rounds 5 shipped= one=$X two=$X@FMD_FIR_DIGITS=2 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --fir --fir-taps-max 127 --rounds 5 two=$X@FMD_FIR_DIGITS=2 one=$X 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --fir --fir-taps-max 127 --out-bufs 4 --rounds 4 one=$X two=$X@FMD_FIR_DIGITS=2 2>/dev/null | sed 's/^/{"out_bufs": 4, /; s/, {/, /' | tee -a $OUT/ab.txt | cut -c1-220
