#!/bin/bash
# session r05be: the stand-alone FIR's two-digit sparse form (re / im split, 16-byte stores): tests, then against the dense form
OUT=gpurun_out/r05be; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_fir.py tests/test_gpu_ref_kat.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.log
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --fir --rounds 5 shipped= sparse=$X dense=$X@FMD_FIR_SPARSE=0 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --fir --rounds 5 dense=$X@FMD_FIR_SPARSE=0 sparse=$X 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --fir --out-bufs 4 --rounds 4 sparse=$X dense=$X@FMD_FIR_SPARSE=0 2>/dev/null | sed 's/^/{"out_bufs": 4, /; s/, {/, /' | tee -a $OUT/ab.txt | cut -c1-220
