#!/bin/bash
# session r04s: is the +0.4 % of the new quotient at downsample >= 10 real?  five more alternating rounds on the HBM-bound configurations
OUT=gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3 4 5; do bash scripts/gpu_ablibs.sh "--cfg 24 --cfg 12,192000,32000 --cfg 14,224000,32000 --cfg 16,150000,32000 --cfg 9,216000,24000" base=libfmd_hip_base.so new=libfmd_hip.so cmulfma=libfmd_hip_cmulfma.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_disc_hbm.txt
python3 tools/ab_summary.py $OUT/ab_disc_hbm.txt
