#!/bin/bash
# session r06r: the streaming kernel's refills as contiguous bursts (variant builds: a contiguous strip of the tile per wave, register
# sets refilled 2 / 4 at a time): full kernel and skeleton against the shipped structure, same process
OUT=gpurun_out/r06r; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/rtl-sdr-rs_amd
FMD_LIB=$L/libfmd_hip_xstripb4.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "4-256000 or 4-200000 or 4-60000" > $OUT/parity_stripb4.log 2>&1; tail -3 $OUT/parity_stripb4.log
timeout 1200 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 4,200000,32000 full=$L/libfmd_hip_exp.so strip=$L/libfmd_hip_xstrip.so batch4=$L/libfmd_hip_xbatch4.so stripb4=$L/libfmd_hip_xstripb4.so stripb2=$L/libfmd_hip_xstripb2.so 2>/dev/null | tee $OUT/ab_full.txt | cut -c1-200
timeout 1200 python tools/ab_libs.py --rounds 3 --cfg 4,256000,48000 --cfg 4,300000,32000 skel=$L/libfmd_hip_exp.so@FMD_DBG=16777216 strip=$L/libfmd_hip_xstrip.so@FMD_DBG=16777216 batch4=$L/libfmd_hip_xbatch4.so@FMD_DBG=16777216 stripb4=$L/libfmd_hip_xstripb4.so@FMD_DBG=16777216 2>/dev/null | tee $OUT/ab_skel.txt | cut -c1-200
