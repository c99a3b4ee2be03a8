#!/bin/bash
# session r04ah: 512-thread blocks (8 waves per tile, 40 KB tiles, 4 tiles per CU) against the shipped 256 (both experiment builds): parity + A/B
OUT=gpurun_out/r04ah; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_t512.so FMD_FUZZ_CASES=60 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 7,166666,32000 --cfg 8,250000,44100 --cfg 12,192000,32000 --cfg 64,37500,8000" t256=libfmd_hip_exp.so t512=libfmd_hip_t512.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_t512.txt
python3 tools/ab_summary.py $OUT/ab_t512.txt
