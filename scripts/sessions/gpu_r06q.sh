#!/bin/bash
# session r06q: the streaming kernel's loads forced to 16-byte alignment (probe, wrong results): what does the misalignment of a tile's
# first window cost?
OUT=gpurun_out/r06q; mkdir -p $OUT; export TMPDIR=/tmp
X=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 900 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 2,500000,32000 full=$X full_aligned=$X@FMD_DBG=33554432 skel=$X@FMD_DBG=16777216 skel_aligned=$X@FMD_DBG=50331648 2>/dev/null | tee $OUT/ab_stream_align.txt | cut -c1-260
