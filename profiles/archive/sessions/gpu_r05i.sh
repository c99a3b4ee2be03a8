#!/bin/bash
# session r05i: pacing probes on the HBM-bound rows (experiment build: s_sleep in front of the DMAs / behind the staging barrier)
OUT=gpurun_out/r05i; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg 24 --cfg 64,37500,8000 --cfg ref r04x=rtl-sdr-rs_amd/libfmd_hip_exp_r04.so base=$X pre128=$X@FMD_DBG=512 pre512=$X@FMD_DBG=1024 post256=$X@FMD_DBG=2048 post1024=$X@FMD_DBG=4096 2>/dev/null | tee $OUT/ab_pace.jsonl | cut -c1-200
