#!/bin/bash
# session r05ak: timelines again on the final experiment build (with and without the tile's output stores), then the round's evidence
OUT=gpurun_out/r05ak; mkdir -p $OUT; export TMPDIR=/tmp
python tools/timeline.py --cfg 24 --cfg ref --cfg 5,250000,44100 2>>$OUT/err.txt | tee -a $OUT/timeline.jsonl | cut -c1-300
python tools/timeline.py --dbg 524288 --cfg 24 --cfg ref 2>>$OUT/err.txt | sed 's/^/{"nostore": 1, /; s/^{"nostore": 1, {/{"nostore": 1, /' | tee -a $OUT/timeline.jsonl | cut -c1-300
bash scripts/gpu_round.sh r05
