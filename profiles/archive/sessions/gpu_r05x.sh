#!/bin/bash
# session r05x: the resampler pass taken apart -- the store alone, the barrier alone, the LDS read alone
OUT=gpurun_out/r05x; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg ref --cfg 5,250000,44100 --cfg 24 base=$X nostore=$X@FMD_DBG=524288 nobar=$X@FMD_DBG=1048576 rscopy=$X@FMD_DBG=4 barestore=$X@FMD_DBG=2097152 barestore_nobar=$X@FMD_DBG=3145728 nors=$X@FMD_DBG=128 nors_nobar=$X@FMD_DBG=1048704 2>/dev/null | tee -a $OUT/ab_rs.txt | cut -c1-200
