#!/bin/bash
# session r04p: fused-bank checkpoint / resume tests; first dot product of every window sum in the three-address form (A/B + parity)
OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_firdemod.py -x -q -m gpu -k "checkpoint or errors" 2>&1 | tail -3 | tee $OUT/pytest_ckpt.log
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_dotinit.so FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest_dotinit.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 12,192000,32000 --cfg 8,250000,44100 --cfg 7,170000,32000 --cfg 5,250000,44100 --cfg 4,300000,50000 --cfg 2,500000,32000" base=libfmd_hip.so new=libfmd_hip_dotinit.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_dotinit.txt
python3 tools/ab_summary.py $OUT/ab_dotinit.txt
