#!/bin/bash
# session r05d: scalar diet step 2 (one-lane regions behind ONE flag test, state loads inside them, resampler dispatch as a decision tree):
# parity, A/B against the round-4 library, instruction mix, per-region counters; the repaired packed-f32 microbenchmark
OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_gpu_ref_kat.py tests/test_gpu_f64_guard.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.log
bash scripts/ab_libs.sh $OUT/ab.txt 3 "--cfg 24 --cfg ref --cfg 4,256000,48000 --cfg 8,250000,44100 --cfg 5,250000,44100 --cfg 7,166666,32000 --cfg 12,192000,32000" r04=rtl-sdr-rs_amd/libfmd_hip_r04.so new= | tee $OUT/ab_summary.txt
: > $OUT/mix.jsonl
bash scripts/pmc_mix.sh $OUT/mix.jsonl "cfg-ref" "cfg-2.4" "D=5" "D=8"
cat $OUT/mix.jsonl
bash scripts/gpu_pmc_regions.sh r05d > $OUT/regions.log 2>&1; tail -12 $OUT/regions.log | cut -c1-330
./tools/valubench > $OUT/valubench.txt 2>&1; tail -24 $OUT/valubench.txt
