#!/bin/bash
# session r05au: what do the stand-alone FIR kernel's tap-fragment loads (5 KB per wave from the L2, 20 KB per 16 KB tile) cost?
OUT=gpurun_out/r05au; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --fir --rounds 4 exp=$X noA=$X@FMD_DBG=16 skeleton=$X@FMD_DBG=5 skeleton_noA=$X@FMD_DBG=21 nostores=$X@FMD_DBG=4 nostores_noA=$X@FMD_DBG=20 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --rounds 3 exp=$X noA=$X@FMD_DBG=16 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
