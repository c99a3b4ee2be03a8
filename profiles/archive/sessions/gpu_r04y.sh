#!/bin/bash
# session r04y: discriminator tail: sign of x as an and + two xors instead of v_bitop3 + v_bfi, the result's low 16 bits by the 1.5 * 2^23 add instead of v_cvt_i32_f32: parity + A/B
OUT=gpurun_out/r04y; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_disct2.so FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 2,500000,32000 --cfg 4,256000,48000 --cfg 5,250000,44100 --cfg 7,166666,32000 --cfg 8,250000,44100 --cfg 12,192000,32000 --cfg 1,48000,48000" base=libfmd_hip_base.so new=libfmd_hip_disct2.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_t2.txt
python3 tools/ab_summary.py $OUT/ab_t2.txt
