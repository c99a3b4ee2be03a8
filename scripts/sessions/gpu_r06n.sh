#!/bin/bash
# session r06n: a long fuzz campaign on the final library (two fresh seeds x 600 cases through every entry point against the oracle)
OUT=gpurun_out/r06n; mkdir -p $OUT; export TMPDIR=/tmp
for seed in 60601 60602; do
  FMD_FUZZ_CASES=600 FMD_FUZZ_SEED=$seed timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -s > $OUT/fuzz_$seed.log 2>&1; echo "seed $seed rc=$?" >> $OUT/summary.txt
  tail -2 $OUT/fuzz_$seed.log >> $OUT/summary.txt
done
