#!/bin/bash
# session r05z: downsample 2 / 4 -- the register-streaming kernel against the LDS tile kernel as it is now (cheaper prologue, row table)
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 2,500000,32000 --cfg 2,96000,48000 stream=$X tile=$X@FMD_STREAM=0 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-250
