#!/bin/bash
# session r05ah: timelines of the looped kernel (experiment build, no spills now): 1 / 2 / 4 tiles per block
OUT=gpurun_out/r05ah; mkdir -p $OUT; export TMPDIR=/tmp
for tpb in 1 2 4; do
python tools/timeline.py --tpb $tpb --cfg 24 --cfg ref --dump $OUT/ref_tpb$tpb.npy 2>>$OUT/err.txt | tee -a $OUT/timeline.jsonl | cut -c1-1000
done
