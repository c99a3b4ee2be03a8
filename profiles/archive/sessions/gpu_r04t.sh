#!/bin/bash
# session r04t: downsample 1 and 3 without the i32-wrap emulation (it cannot engage there): parity + timing against the pre-r04r library
OUT=gpurun_out/r04t; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 1,48000,48000 --cfg 3,150000,48000 --cfg 2,500000,32000 --cfg 5,250000,44100" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_nowrap.txt
python3 tools/ab_summary.py $OUT/ab_nowrap.txt
