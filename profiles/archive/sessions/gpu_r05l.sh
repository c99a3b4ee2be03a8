#!/bin/bash
# session r05l: tile-count sweep at the headline and its neighbours (experiment build, FMD_KT), with and without the pacing sleep
OUT=gpurun_out/r05l; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
V="base=$X"; for k in 100 104 106 108 110 112 114 116 120 122 124 126 128 130 134 138; do V="$V k$k=$X@FMD_KT=$k"; done
python tools/ab_libs.py --rounds 3 --cfg 24 $V 2>/dev/null | tee $OUT/kt24.jsonl | cut -c1-210
V="base=$X rs=$X@FMD_DBG=32768"; for k in 110 114 126 134; do V="$V k$k=$X@FMD_KT=$k k${k}rs=$X@FMD_KT=$k,FMD_DBG=32768"; done
python tools/ab_libs.py --rounds 3 --cfg 24 $V 2>/dev/null | tee $OUT/kt24rs.jsonl | cut -c1-210
python tools/ab_libs.py --rounds 3 --cfg 12,192000,32000 --cfg 16,150000,32000 --cfg 64,37500,8000 --cfg 9,180000,32000 --cfg 11,220000,32000 --cfg 13,208000,32000 --cfg 14,224000,32000 --cfg 7,166666,32000 base=$X rs=$X@FMD_DBG=32768 post=$X@FMD_DBG=4096 2>/dev/null | tee $OUT/rs_domain.jsonl | cut -c1-210
