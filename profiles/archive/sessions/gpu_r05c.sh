#!/bin/bash
# session r05c: the rich tile rows (whole tile context from the host-built table): parity, A/B against the round-4 library, SALU per wave
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_gpu_ref_kat.py tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.log
bash scripts/ab_libs.sh $OUT/ab.txt 3 "--cfg 24 --cfg ref --cfg 4,256000,48000 --cfg 8,250000,44100 --cfg 5,250000,44100 --cfg 2,500000,32000" r04=rtl-sdr-rs_amd/libfmd_hip_r04.so new= | tee $OUT/ab_summary.txt
: > $OUT/mix.jsonl
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_r04.so bash scripts/pmc_mix.sh $OUT/mix.jsonl "cfg-ref" "cfg-2.4"
bash scripts/pmc_mix.sh $OUT/mix.jsonl "cfg-ref" "cfg-2.4"
cat $OUT/mix.jsonl
