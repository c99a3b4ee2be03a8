#!/bin/bash
# session r04q: masked-window rounds (odd downsample): pointer stepping, full / tail rounds, dead-register DPP, three-address dot2: parity + A/B
OUT=gpurun_out/r04q; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 7,170000,32000 --cfg 5,250000,44100 --cfg 3,150000,48000 --cfg 9,216000,24000 --cfg 11,264000,24000 --cfg 14,224000,32000 --cfg 15,240000,16000 --cfg ref --cfg 24" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_masked.txt
python3 tools/ab_summary.py $OUT/ab_masked.txt
