#!/bin/bash
# session r05k: is the pacing / tile-count effect real?  the unchanged build at three positions of every round
OUT=gpurun_out/r05k; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg 24 --cfg 64,37500,8000 base=$X post1024=$X@FMD_DBG=4096 base2=$X rs1024=$X@FMD_DBG=32768 base3=$X k114=$X@FMD_KT=114 k126=$X@FMD_KT=126 base4=$X 2>/dev/null | tee $OUT/ab_pos.jsonl | cut -c1-200
python tools/ab_libs.py --rounds 4 --cfg 24 --cfg 64,37500,8000 --cfg ref s3= r04=rtl-sdr-rs_amd/libfmd_hip_r04.so s3b= r04b=rtl-sdr-rs_amd/libfmd_hip_r04.so 2>/dev/null | tee $OUT/ab_pos2.jsonl | cut -c1-200
