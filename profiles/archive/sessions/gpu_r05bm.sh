#!/bin/bash
# session r05bm: the three fuzz tests at ten times their default case counts on the final library (two seeds each)
OUT=gpurun_out/r05bm; mkdir -p $OUT; export TMPDIR=/tmp
for seed in 11 12; do
FMD_FUZZ_CASES=300 FMD_FUZZ_SEED=$seed timeout 2400 python -m pytest tests/test_fir.py::test_gpu_fir_fuzz tests/test_firdemod.py::test_gpu_fused_fuzz tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 | sed "s/^/seed=$seed /" | tee -a $OUT/fuzz.log
done
