#!/bin/bash
# session r04x: steady-state (clock-settled, base-bracketed) instruction costs; the i32 wrap as fma / rndne / fma instead of four adds: parity + A/B
OUT=gpurun_out/r04x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 tools/pkbench > $OUT/pkbench.txt 2>&1
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_wrapfma.so FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 4,256000,48000 --cfg 5,250000,44100 --cfg 7,166666,32000 --cfg 8,250000,44100 --cfg 12,192000,32000" base=libfmd_hip_base.so new=libfmd_hip_wrapfma.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_wrap.txt
python3 tools/ab_summary.py $OUT/ab_wrap.txt
awk '{print $1, $2, $(NF-7), $(NF-6), $NF}' $OUT/pkbench.txt | sed -n '1~2p'
