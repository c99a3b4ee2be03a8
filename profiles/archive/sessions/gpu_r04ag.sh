#!/bin/bash
# session r04ag: odd downsample 3 ... 15 in the adjacent-window form (f32 components, per-lane weights): parity + A/B
OUT=gpurun_out/r04ag; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=200 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py tests/test_gpu_sink.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 5,250000,44100 --cfg 7,166666,32000 --cfg 3,150000,48000 --cfg 9,216000,24000 --cfg 11,264000,24000 --cfg 13,208000,32000 --cfg 15,240000,16000 --cfg 5,240000,32000" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_odd.txt
python3 tools/ab_summary.py $OUT/ab_odd.txt
