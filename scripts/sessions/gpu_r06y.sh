#!/bin/bash
# session r06y: what is left in the one-shot streaming kernel's skeleton (bit 24): without the audio stores (bit 19), without the
# resampler pass altogether (bit 7), without the block barrier (bit 20)
OUT=gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp
X=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 900 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 full=$X skel=$X@FMD_DBG=$((1<<24)) skel_nostore=$X@FMD_DBG=$(((1<<24)|(1<<19))) skel_nopass=$X@FMD_DBG=$(((1<<24)|(1<<7))) skel_nopass_nobarrier=$X@FMD_DBG=$(((1<<24)|(1<<7)|(1<<20))) full_nopass=$X@FMD_DBG=$((1<<7)) 2>/dev/null | tee $OUT/ab.txt | cut -c1-220
timeout 300 python tools/ab_libs.py --rounds 3 --cfg 24 --cfg ref full=$X staging_skel=$X@FMD_DBG=8 2>/dev/null | tee $OUT/ab_tile.txt | cut -c1-220
