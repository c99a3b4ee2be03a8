#!/bin/bash
# session r06p: the streaming kernel's skeleton (experiment build, ablation bit 24: loads and one store per round, no arithmetic) -- does
# its load pattern alone reach the LDS-DMA kernel's staging skeleton (FMD_DBG=8)?
OUT=gpurun_out/r06p; mkdir -p $OUT; export TMPDIR=/tmp
X=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 900 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 2,500000,32000 full=$X skel=$X@FMD_DBG=16777216 skel_noresample=$X@FMD_DBG=16777220 noresample=$X@FMD_DBG=4 2>/dev/null | tee $OUT/ab_stream_skel.txt | cut -c1-260
timeout 600 python tools/ab_libs.py --rounds 4 --cfg 24 --cfg ref --cfg 8,250000,44100 full=$X staging_skel=$X@FMD_DBG=8 2>/dev/null | tee $OUT/ab_tile_skel.txt | cut -c1-260
