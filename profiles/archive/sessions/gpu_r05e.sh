#!/bin/bash
# session r05e: same-process A/B of three builds (round 4, scalar diet step 1, step 2) -- r05d's alternating processes disagreed with r05c's
OUT=gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_firdemod.py tests/test_fir.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.log
python tools/ab_libs.py --rounds 5 --cfg 24 --cfg ref --cfg 4,256000,48000 --cfg 8,250000,44100 --cfg 5,250000,44100 --cfg 12,192000,32000 --cfg 64,37500,8000 r04=rtl-sdr-rs_amd/libfmd_hip_r04.so s1=rtl-sdr-rs_amd/libfmd_hip_s1.so s2= 2>/dev/null | tee $OUT/ab3.jsonl | cut -c1-260
python tools/ab_libs.py --rounds 5 --cfg 24 --cfg ref s2= s1=rtl-sdr-rs_amd/libfmd_hip_s1.so r04=rtl-sdr-rs_amd/libfmd_hip_r04.so 2>/dev/null | tee $OUT/ab3_rev.jsonl | cut -c1-260
