#!/bin/bash
# one bench line per call (box spread): appended locally to profiles/r05_box_spread.jsonl
OUT=gpurun_out/r05r; mkdir -p $OUT
python bench.py --no-cpu 2>/dev/null > $OUT/bench_$(date +%s).json
