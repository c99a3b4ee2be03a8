#!/bin/bash
# session r04w: resampler group sums by v_dot2 over the half-word pairs as they lie (one unaligned read, as before): parity + A/B
OUT=gpurun_out/r04w; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_gsdot2.so FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 1,48000,48000 --cfg 4,256000,48000 --cfg 5,250000,44100 --cfg 7,166666,32000 --cfg 8,250000,44100 --cfg 3,150000,48000 --cfg 12,192000,32000" base=libfmd_hip_base.so new=libfmd_hip_gsdot2.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_gs.txt
python3 tools/ab_summary.py $OUT/ab_gs.txt
