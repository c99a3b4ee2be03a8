#!/bin/bash
# session r04ad: PROBE (results wrong on purpose): how much would starting the rounds before the tile's last DMAs have landed buy?
OUT=gpurun_out/r04ad; mkdir -p $OUT; export TMPDIR=/tmp
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for cfg in ref 24 "5,250000,44100" "7,166666,32000" "8,250000,44100"; do
  python tools/ab.py --rounds 3 --cfg $cfg full: early2:FMD_DBG=256 early1:FMD_DBG=512 2>/dev/null | grep '^{"cfg"'
done > $OUT/ab_probe.txt
unset FMD_LIB
python3 tools/ab_summary.py $OUT/ab_probe.txt
