#!/bin/bash
# session r06z: what the audio stores of the one-shot streaming kernel cost (downsample 4: 4.7 % of the bytes, 18 % of the time) --
# everything but the store (bit 19), stores that stay in the L2 (22), nontemporal (18), the bare store (21), an explicit wait (23)
OUT=gpurun_out/r06z; mkdir -p $OUT; export TMPDIR=/tmp
X=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 900 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 4,200000,32000 full=$X nostore=$X@FMD_DBG=$((1<<19)) l2only=$X@FMD_DBG=$((1<<22)) nt=$X@FMD_DBG=$((1<<18)) barestore=$X@FMD_DBG=$((1<<21)) wait=$X@FMD_DBG=$((1<<23)) 2>/dev/null | tee $OUT/ab.txt | cut -c1-220
timeout 600 python tools/ab_libs.py --rounds 3 --cap-align 64 --cfg 4,256000,48000 --cfg 4,300000,32000 full_aligned_rows=$X nostore=$X@FMD_DBG=$((1<<19)) 2>/dev/null | tee $OUT/ab_align.txt | cut -c1-220
