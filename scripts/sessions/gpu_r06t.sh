#!/bin/bash
# session r06t: one-shot waves in the streaming kernel (variant: 8 rounds per wave, every load issued up front, no refills -- the shape
# of tools/membench's tile kernels, which read 6.3 - 7.0 TB/s where the grid-stride stream reads 5.3 - 5.6)
OUT=gpurun_out/r06t; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/rtl-sdr-rs_amd
FMD_LIB=$L/libfmd_hip_xone8.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "4-256000 or 4-200000 or 4-60000 or 2-500000 or 2-1000000" > $OUT/parity_one8.log 2>&1; tail -3 $OUT/parity_one8.log
timeout 1200 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 4,200000,32000 --cfg 2,500000,32000 full=$L/libfmd_hip_exp.so one8=$L/libfmd_hip_xone8.so skel=$L/libfmd_hip_exp.so@FMD_DBG=16777216 one8skel=$L/libfmd_hip_xone8.so@FMD_DBG=16777216 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
