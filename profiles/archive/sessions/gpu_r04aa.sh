#!/bin/bash
# session r04aa: rate_out == rate_resample as a copy instead of a resampler pass: parity (incl. the new equal-rate configurations) + A/B
OUT=gpurun_out/r04aa; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py tests/test_gpu_sink.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 1,48000,48000 --cfg 4,60000,60000 --cfg 10,24000,24000 --cfg ref --cfg 24" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_copy.txt
python3 tools/ab_summary.py $OUT/ab_copy.txt
