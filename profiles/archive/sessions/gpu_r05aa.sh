#!/bin/bash
# session r05aa: the stand-alone FIR kernel by region (ablation bits 0 - 3 of the experiment build), same process
OUT=gpurun_out/r05aa; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --fir --rounds 4 shipped= exp=$X nomfma=$X@FMD_DBG=1 nostores=$X@FMD_DBG=4 skeleton=$X@FMD_DBG=5 noloads=$X@FMD_DBG=2 nohist=$X@FMD_DBG=8 2>/dev/null | tee -a $OUT/ab_fir.txt | cut -c1-250
