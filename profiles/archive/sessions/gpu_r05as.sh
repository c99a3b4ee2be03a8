#!/bin/bash
# session r05as: the fused FIR kernel's column length again, now that the tile budgets are whole LDS granules (NG 6 / 7: 7 / 6 blocks per CU)
OUT=gpurun_out/r05as; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --firdemod --rounds 4 shipped= ng8=$X ng7=$X@FMD_FD_REG=7 ng6=$X@FMD_FD_REG=6 ng5=$X@FMD_FD_REG=5 ng10=$X@FMD_FD_REG=10 ng7b=$X@FMD_FD_REG=7,FMD_FD_LDS=32000 ng6b=$X@FMD_FD_REG=6,FMD_FD_LDS=26880 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
