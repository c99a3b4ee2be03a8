#!/bin/bash
# session r05an: split staging (the first rounds under the rest of the tile's load): parity on the variant build, then same-process A/B
OUT=gpurun_out/r05an; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
FMD_LIB=$PWD/$P/libfmd_hip_ss.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ref_kat.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT/parity.txt
python tools/ab_libs.py --rounds 4 --cfg ref --cfg 24 --cfg 8,250000,44100 --cfg 12,192000,32000 --cfg 4,256000,48000 --cfg 14,224000,32000 --cfg 6,200000,48000 --cfg 10,250000,48000 base= ss=$P/libfmd_hip_ss.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-200
