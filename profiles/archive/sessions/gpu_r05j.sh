#!/bin/bash
# session r05j: pacing probes, second pass (longer sleeps behind the barrier, a sleep in front of the resampler) + tile sizes on the new kernels
OUT=gpurun_out/r05j; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg 24 --cfg 64,37500,8000 --cfg 12,192000,32000 --cfg 16,150000,32000 --cfg 8,250000,44100 base=$X post1024=$X@FMD_DBG=4096 post2048=$X@FMD_DBG=8192 post4096=$X@FMD_DBG=16384 post3072=$X@FMD_DBG=12288 rs1024=$X@FMD_DBG=32768 both=$X@FMD_DBG=36864 2>/dev/null | tee $OUT/ab_pace2.jsonl | cut -c1-200
python tools/ab_libs.py --rounds 3 --cfg 24 base=$X k110=$X@FMD_KT=110 k114=$X@FMD_KT=114 k122=$X@FMD_KT=122 k126=$X@FMD_KT=126 k134=$X@FMD_KT=134 2>/dev/null | tee $OUT/ab_kt24.jsonl | cut -c1-200
