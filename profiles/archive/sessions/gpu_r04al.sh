#!/bin/bash
# session r04al: scalar diet, step 4 (even boxcar phase a compile-time fact of the even factors' fast kernels): parity + A/B
OUT=gpurun_out/r04al; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 8,250000,44100 --cfg 12,192000,32000 --cfg 64,37500,8000 --cfg 4,256000,48000 --cfg 16,150000,32000" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_salu4.txt
python3 tools/ab_summary.py $OUT/ab_salu4.txt
