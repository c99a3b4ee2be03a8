#!/bin/bash
# session r05y: is the store's cost its HBM write or the wait for its acknowledgement?
OUT=gpurun_out/r05y; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg ref --cfg 5,250000,44100 --cfg 24 base=$X nostore=$X@FMD_DBG=524288 l2store=$X@FMD_DBG=4194304 waitstore=$X@FMD_DBG=8388608 l2wait=$X@FMD_DBG=12582912 2>/dev/null | tee -a $OUT/ab_rs.txt | cut -c1-200
