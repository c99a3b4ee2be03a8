#!/bin/bash
# session r05af: several consecutive tiles per block (FMD_TPB, experiment build): parity, same-process A/B, timeline
OUT=gpurun_out/r05af; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
for tpb in 2 3; do
FMD_LIB=$PWD/$X FMD_TPB=$tpb timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ref_kat.py -x -q -m gpu 2>&1 | tail -3 | sed "s/^/tpb=$tpb /" | tee -a $OUT/parity.txt
done
python tools/ab_libs.py --rounds 3 --cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 8,250000,44100 --cfg 7,166666,32000 --cfg 64,37500,8000 --cfg 3,400000,48000 --cfg 1,48000,48000 shipped= tpb1=$X tpb2=$X@FMD_TPB=2 tpb3=$X@FMD_TPB=3 tpb4=$X@FMD_TPB=4 tpb6=$X@FMD_TPB=6 tpb9=$X@FMD_TPB=9 tpb32=$X@FMD_TPB=32 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-200
for tpb in 2 4; do
python tools/timeline.py --tpb $tpb --cfg 24 --cfg ref 2>>$OUT/err.txt | tee -a $OUT/timeline.jsonl | cut -c1-900
done
