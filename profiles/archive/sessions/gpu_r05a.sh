#!/bin/bash
# session r05a: the reference's KAT vectors through the HIP path + a baseline bench line of the round-4 kernels on this round's first box
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ref_kat.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/kat.log
timeout 600 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-600
