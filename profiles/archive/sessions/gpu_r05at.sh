#!/bin/bash
# session r05at: the fused FIR kernel with the corrected whole-granule budgets: NG 6 / 7 / 8 at config 4, then its GPU tests
OUT=gpurun_out/r05at; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --firdemod --rounds 3 shipped= ng8=$X ng7=$X@FMD_FD_REG=7 ng6=$X@FMD_FD_REG=6 ng7old=$X@FMD_FD_REG=7,FMD_FD_LDS=27300 ng6old=$X@FMD_FD_REG=6,FMD_FD_LDS=23400 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
timeout 1200 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
