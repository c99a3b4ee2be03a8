#!/bin/bash
# session r05ao: does the XCDs' relative position in memory matter?  (channels per XCD region 496 ... 520: region bases 124 ... 130 MB apart)
OUT=gpurun_out/r05ao; mkdir -p $OUT; export TMPDIR=/tmp
for ch in 4096 4104 4160 3968 4032 4096; do
python tools/ab_libs.py --rounds 3 --channels $ch --cfg 24 --cfg ref shipped= 2>/dev/null | sed "s/^/{\"channels\": $ch, /; s/, {/, /" | tee -a $OUT/ab.txt | cut -c1-200
done
