#!/bin/bash
# session r06u: one-shot waves in the streaming kernel, 4 / 6 / 7 / 8 rounds per wave at downsample 4 and 8 / 12 / 14 / 16 at downsample 2
# (variant builds of the experiment library) against the library of commit 4ae6769 (12 / 16 rounds, 8 loads in flight, refills)
OUT=gpurun_out/r06u; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/rtl-sdr-rs_amd
for v in x48 x46; do
FMD_LIB=$L/libfmd_hip_$v.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "4-256000 or 4-200000 or 4-60000 or 2-500000 or 2-1000000" > $OUT/parity_$v.log 2>&1; tail -1 $OUT/parity_$v.log
done
timeout 1500 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,300000,32000 --cfg 4,200000,32000 --cfg 4,250000,44100 --cfg 2,500000,32000 --cfg 2,500000,48000 base=$L/libfmd_hip_r06baseexp.so r8_16=$L/libfmd_hip_x48.so r7_14=$L/libfmd_hip_x47.so r6_12=$L/libfmd_hip_x46.so r4_8=$L/libfmd_hip_x44.so 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
