#!/bin/bash
# session r06ab: the streaming kernels' fresh waves at issue priority 3 until their loads are out (as the LDS-DMA kernels do)
OUT=gpurun_out/r06ab; mkdir -p $OUT; export TMPDIR=/tmp
BASE=$PWD/rtl-sdr-rs_amd/libfmd_hip_r06b.so
timeout 900 python tools/ab_libs.py --rounds 5 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 4,250000,44100 --cfg 2,500000,32000 --cfg 2,500000,48000 base=$BASE new= 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
