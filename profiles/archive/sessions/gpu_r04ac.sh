#!/bin/bash
# session r04ac: register-streaming rounds: compile-time store offsets + one compare per store, v_med3 clamp, constants pinned in registers: parity + A/B
OUT=gpurun_out/r04ac; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_streamt.so FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 2,500000,32000 --cfg 4,256000,48000 --cfg 4,300000,50000 --cfg 2,64000,32000 --cfg 4,192000,48000" base=libfmd_hip_base.so new=libfmd_hip_streamt.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_stream.txt
python3 tools/ab_summary.py $OUT/ab_stream.txt
