#!/bin/bash
# session r04ai: PROBE: what do 64 / 128 extra scalar instructions per wave cost, against 64 extra vector ones?  (tools/salubench: a
# scalar add costs more issue time per SIMD than a vector add)
OUT=gpurun_out/r04ai; mkdir -p $OUT; export TMPDIR=/tmp
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for cfg in ref 24 "8,250000,44100" "64,37500,8000"; do
  python tools/ab.py --rounds 3 --cfg $cfg full: s64:FMD_DBG=1024 s128:FMD_DBG=2048 v64:FMD_DBG=4096 2>/dev/null | grep '^{"cfg"'
done > $OUT/ab_salu.txt
unset FMD_LIB
python3 tools/ab_summary.py $OUT/ab_salu.txt
