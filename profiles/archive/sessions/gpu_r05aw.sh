#!/bin/bash
# session r05aw: the one-digit form of the fused FIR kernel: its tests, then 8-bit taps through one and two digits side by side
OUT=gpurun_out/r05aw; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_firdemod.py tests/test_gpu_f64_guard.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.log
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --firdemod --fir-taps-max 127 --rounds 4 shipped= one=$X two=$X@FMD_FD_DIGITS=2 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --rounds 3 shipped= two=$X 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
