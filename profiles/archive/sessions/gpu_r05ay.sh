#!/bin/bash
# session r05ay: the fused FIR kernel's matrix phase on the 4:2 sparse matrix instruction: tests, then sparse against dense side by side
OUT=gpurun_out/r05ay; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_firdemod.py tests/test_gpu_f64_guard.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.log
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --firdemod --rounds 4 shipped= sparse=$X dense=$X@FMD_FD_SPARSE=0 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --fir-taps-max 127 --rounds 4 shipped= sparse=$X dense=$X@FMD_FD_SPARSE=0 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
