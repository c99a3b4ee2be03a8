#!/bin/bash
# session r05h: scalar diet step 3 -- ONE scalar-memory round trip per block (geometry block + whole row in front of the DMAs, the
# resampler's parameters under the DMAs' latency): parity; A/B round 4 / step 2 / step 3 in one process; instruction mix
OUT=gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_gpu_ref_kat.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.log
python tools/ab_libs.py --rounds 5 --cfg 24 --cfg ref --cfg 64,37500,8000 --cfg 12,192000,32000 --cfg 5,250000,44100 --cfg 8,250000,44100 --cfg 4,256000,48000 --cfg 2,500000,32000 --cfg 1,48000,48000 r04=rtl-sdr-rs_amd/libfmd_hip_r04.so s2=rtl-sdr-rs_amd/libfmd_hip_s2.so s3= 2>/dev/null | tee $OUT/ab3.jsonl | cut -c1-230
python tools/ab_libs.py --rounds 5 --cfg 24 --cfg ref --cfg 64,37500,8000 s3= r04=rtl-sdr-rs_amd/libfmd_hip_r04.so 2>/dev/null | tee $OUT/ab3_rev.jsonl | cut -c1-230
: > $OUT/mix.jsonl
bash scripts/pmc_mix.sh $OUT/mix.jsonl "cfg-ref" "cfg-2.4" "D=64"
cat $OUT/mix.jsonl | cut -c1-330
# priority knobs of the fused FIR kernel (experiment build)
python tools/ab_libs.py --firdemod --rounds 5 base=rtl-sdr-rs_amd/libfmd_hip_exp.so mfma3=rtl-sdr-rs_amd/libfmd_hip_exp.so@FMD_DBG=16777216 mfma1=rtl-sdr-rs_amd/libfmd_hip_exp.so@FMD_DBG=33554432 disc2=rtl-sdr-rs_amd/libfmd_hip_exp.so@FMD_DBG=67108864 2>/dev/null | tee $OUT/ab_fd_prio.jsonl | cut -c1-200
