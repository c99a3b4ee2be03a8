#!/bin/bash
# session r05ab: the stand-alone FIR's staging skeleton against the call length (is the 0.82 of its skeleton the launch's ramp and tail?)
OUT=gpurun_out/r05ab; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
for nb in 1048576 2097152 8388608; do
python tools/ab_libs.py --fir --fir-bytes $nb --rounds 3 --steps 50 --settle 30 exp=$X nostores=$X@FMD_DBG=4 skeleton=$X@FMD_DBG=5 2>/dev/null | tee -a $OUT/ab_fir.txt | cut -c1-250
done
python tools/ab_libs.py --fir --fir-channels 1024 --fir-bytes 2097152 --rounds 3 --steps 50 --settle 30 exp=$X nostores=$X@FMD_DBG=4 skeleton=$X@FMD_DBG=5 2>/dev/null | tee -a $OUT/ab_fir.txt | cut -c1-250
