#!/bin/bash
# session r06k: per-launch kernel durations and gaps of the four cadences (which launch of every fifteen carries the bubble?)
OUT=gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/pt
timeout 300 rocprofv3 --kernel-trace -d $OUT/pt -o pt -f csv -- python3 tools/pipelined_trace.py > $OUT/trace.json 2> $OUT/pt.err
python3 tools/pipelined_gaps.py $OUT/pt > $OUT/gaps.json
rm -rf $OUT/pt
