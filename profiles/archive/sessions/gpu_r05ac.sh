#!/bin/bash
# session r05ac: one output buffer written again by every call (the bench's form) against outputs rotating over more than the memory-side cache holds
OUT=gpurun_out/r05ac; mkdir -p $OUT; export TMPDIR=/tmp
for nb in 1 4; do
python tools/ab_libs.py --fir --out-bufs $nb --rounds 3 shipped= 2>/dev/null | sed "s/^/outbufs=$nb /" | tee -a $OUT/ab.txt | cut -c1-250
done
for nb in 1 12; do
python tools/ab_libs.py --out-bufs $nb --rounds 3 --cfg 24 --cfg ref shipped= 2>/dev/null | sed "s/^/outbufs=$nb /" | tee -a $OUT/ab.txt | cut -c1-250
done
