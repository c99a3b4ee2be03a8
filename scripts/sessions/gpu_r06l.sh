#!/bin/bash
# session r06l: the light settle path (guarded samples settled without draining the queue): guard tests, then the four cadences again
OUT=gpurun_out/r06l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_f64_guard.py tests/test_gpu_boundary.py -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
rm -rf $OUT/pt
timeout 300 rocprofv3 --kernel-trace -d $OUT/pt -o pt -f csv -- python3 tools/pipelined_trace.py > $OUT/trace.json 2> $OUT/pt.err
python3 tools/pipelined_gaps.py $OUT/pt > $OUT/gaps.json
rm -rf $OUT/pt
timeout 300 python3 bench.py --steps 300 --warmup 50 > $OUT/bench.json 2> $OUT/bench.err
