#!/bin/bash
# session r05bc: the two-digit sparse form with re / im in accumulators of their own: tests, then sparse (shipped) against dense builds side by side
OUT=gpurun_out/r05bc; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
timeout 2400 python -m pytest tests/test_firdemod.py tests/test_gpu_f64_guard.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest.log
python tools/ab_libs.py --firdemod --rounds 5 sparse= dense=$P/libfmd_hip_dn.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --rounds 5 dense=$P/libfmd_hip_dn.so sparse= 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --fir-taps-max 127 --rounds 4 sparse= dense=$P/libfmd_hip_dn.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
