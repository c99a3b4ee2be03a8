#!/bin/bash
# session r05ae: timeline of one launch (probe of the experiment build): resident tiles per CU, staging / compute share of a tile's life, slot turn-over
OUT=gpurun_out/r05ae; mkdir -p $OUT; export TMPDIR=/tmp
python tools/timeline.py --cfg 24 --cfg ref --cfg 5,250000,44100 --cfg 64,37500,8000 --dump $OUT/last.npy 2>$OUT/err.txt | tee $OUT/timeline.jsonl | cut -c1-1200
tail -3 $OUT/err.txt
