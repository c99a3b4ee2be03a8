#!/bin/bash
# session r05v: what would a free resampler / free rounds buy at the reference's rates?  (ablation bits of the experiment build, un-profiled times)
OUT=gpurun_out/r05v; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --rounds 4 --cfg ref --cfg 5,250000,44100 --cfg 24 base=$X nors=$X@FMD_DBG=128 rscopy=$X@FMD_DBG=4 nodisc=$X@FMD_DBG=1 norounds=$X@FMD_DBG=64 skel=$X@FMD_DBG=8 2>/dev/null | tee $OUT/ab_ablate.jsonl | cut -c1-200
