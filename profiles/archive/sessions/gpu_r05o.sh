#!/bin/bash
# session r05o: pacing on top of the two-budget tilings (experiment build): 256 / 512 / 1024 clocks in front of the resampler, 256 / 1024 behind the barrier
OUT=gpurun_out/r05o; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 3 --cfg 24 --cfg 12,192000,32000 --cfg 14,224000,32000 --cfg 15,240000,32000 --cfg 16,150000,32000 --cfg 32,512000,32000 --cfg 64,37500,8000 --cfg 20,200000,48000 --cfg 11,220000,32000 --cfg 13,208000,32000 --cfg 10,250000,48000 --cfg 9,180000,32000 base=$X rs256=$X@FMD_DBG=131072 rs512=$X@FMD_DBG=65536 rs1024=$X@FMD_DBG=32768 post256=$X@FMD_DBG=2048 post1024=$X@FMD_DBG=4096 base2=$X 2>/dev/null | tee $OUT/ab_pace3.jsonl | cut -c1-200
