#!/bin/bash
# session r04ak: scalar diet, step 3 (the resampler's group-length switch in front of the loop instead of inside it): parity + A/B
OUT=gpurun_out/r04ak; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 8,250000,44100 --cfg 64,37500,8000 --cfg 4,256000,48000 --cfg 7,166666,32000" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_salu3.txt
python3 tools/ab_summary.py $OUT/ab_salu3.txt
