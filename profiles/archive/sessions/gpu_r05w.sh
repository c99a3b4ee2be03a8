#!/bin/bash
# session r05w: what do the output stores cost, and can their policy / the row alignment change it?
OUT=gpurun_out/r05w; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
for al in 1 64 256; do
python tools/ab_libs.py --rounds 4 --cap-align $al --cfg ref --cfg 24 --cfg 5,250000,44100 base=$X nt=$X@FMD_DBG=262144 nors=$X@FMD_DBG=128 2>/dev/null | sed "s/^/align=$al /" | tee -a $OUT/ab_store.txt | cut -c1-200
done
