#!/bin/bash
# session r05s: 128-thread blocks (two waves per tile, 16 tiles of ~10 KB per CU) on the round-5 kernels -- round 3 measured them +3 ... 9 %
# slower, with the per-tile costs of that round; scratch build (/tmp copy: FMD_BLOCK_THREADS 128, LDS budgets halved)
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_b128.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config1 or batched_random or config3 or phase_classes" 2>&1 | tail -4 | tee $OUT/pytest_b128.log
python tools/ab_libs.py --rounds 4 --cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 8,250000,44100 --cfg 7,166666,32000 --cfg 12,192000,32000 --cfg 3,250000,48000 s4= b128=rtl-sdr-rs_amd/libfmd_hip_b128.so 2>/dev/null | tee $OUT/ab_b128.jsonl | cut -c1-220
