#!/bin/bash
# session r06x: downsample 2 / 4 through the LDS-DMA tile kernels (FMD_STREAM=0, experiment build) against the streaming kernels as shipped
OUT=gpurun_out/r06x; mkdir -p $OUT; export TMPDIR=/tmp
X=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 900 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 2,500000,32000 stream=$X dma=$X@FMD_STREAM=0 2>/dev/null | tee $OUT/ab.txt | cut -c1-220
