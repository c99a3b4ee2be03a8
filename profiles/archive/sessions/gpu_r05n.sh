#!/bin/bash
# session r05n: the two-budget tiling choice (memory-side rows on ~17 KB tiles) against the previous build (s3: 20 KB everywhere) and round 4;
# pacing probes of the FIR kernels; parity on the new tilings
OUT=gpurun_out/r05n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.log
python tools/ab_libs.py --rounds 4 --cfg 24 --cfg 12,192000,32000 --cfg 16,150000,32000 --cfg 11,220000,32000 --cfg 13,208000,32000 --cfg 15,240000,32000 --cfg 20,200000,48000 --cfg 32,512000,32000 --cfg 64,37500,8000 --cfg 14,224000,32000 --cfg ref --cfg 10,250000,48000 r04=rtl-sdr-rs_amd/libfmd_hip_r04.so s3=rtl-sdr-rs_amd/libfmd_hip_s3.so s4= 2>/dev/null | tee $OUT/ab_tiling.jsonl | cut -c1-230
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab_libs.py --firdemod --rounds 4 base=$X post=$X@FMD_DBG=134217728 rs=$X@FMD_DBG=268435456 2>/dev/null | tee $OUT/ab_fd_pace.jsonl | cut -c1-200
for d in 0 256 512 1024; do FMD_LIB=$PWD/$X FMD_DBG=$d python tools/bench_fir.py 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/FIR dbg=$d /" | tee -a $OUT/fir_pace.txt; done
for d in 0 256 512 1024; do FMD_LIB=$PWD/$X FMD_DBG=$d python tools/bench_fir.py 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/FIR dbg=$d /" | tee -a $OUT/fir_pace.txt; done
