#!/bin/bash
# session r06m: fmd_demod_check without the head copy (a launch that reports sets a host-mapped word): guard / boundary / sink tests,
# the cost of the completion point in pieces (tools/check_cost.py), the bench line
OUT=gpurun_out/r06m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_f64_guard.py tests/test_gpu_boundary.py tests/test_gpu_parity.py -q -x -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
python3 tools/check_cost.py > $OUT/check_cost_flag.json 2> $OUT/err.log
timeout 300 python3 bench.py --steps 300 --warmup 50 > $OUT/bench.json 2> $OUT/bench.err
